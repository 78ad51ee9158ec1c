// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// See pic_kernels.hpp for the statement of purpose and the parity pin.
//
// This file: (1) kernel-level C entry points `orc_*` with the same signatures as
// the product C-ABI (include/warpx_amd.h) but host pointers; (2) the
// single-level periodic step schedule (`orc_sim_*`) restating
// WarpX::Evolve / OneStep_nosub (Source/Evolve/WarpXEvolve.cpp:94-347,354-455);
// (3) the diagnostics formulas that define the parity metric and the golden
// checksums (FieldEnergy, ParticleEnergy, ParticleMomentum, cell-centred sum|Q|).
#include "pic_kernels.hpp"
#include "nci_godfrey_tables.hpp"

#include <climits>
#include <array>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

inline int vlo(const wxa_field_view& f, int d) { return f.lo[d] + f.ng[d]; }
inline int vhi(const wxa_field_view& f, int d) { return f.lo[d] + f.n[d] - f.ng[d]; }  // exclusive
inline int ncell_of(const wxa_field_view& f, int d) { return f.n[d] - 2 * f.ng[d] - f.stag[d]; }
inline int clo_(const wxa_field_view& f, int d, const int32_t* c) { return c ? std::max(vlo(f, d), (int)c[d]) : vlo(f, d); }
inline int chi_(const wxa_field_view& f, int d, const int32_t* c) { return c ? std::min(vhi(f, d), (int)c[d]) : vhi(f, d); }

int max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// CartesianNodalAlgorithm::UpwardD{x,y,z} == DownwardD{x,y,z} (CartesianNodalAlgorithm.H:71-180): centred
// differences on the collocated grid (warpx.grid_type = collocated)
struct NodalD {
    double idx, idy, idz;
    double Dx(const Arr& F, int i, int j, int k) const { return 0.5 * idx * (F(i + 1, j, k) - F(i - 1, j, k)); }
    double Dy(const Arr& F, int i, int j, int k) const { return 0.5 * idy * (F(i, j + 1, k) - F(i, j - 1, k)); }
    double Dz(const Arr& F, int i, int j, int k) const { return 0.5 * idz * (F(i, j, k + 1) - F(i, j, k - 1)); }
};

inline bool all_nodal(const wxa_field_view F[3]) {
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d)
            if (!F[c].stag[d]) return false;
    return true;
}

// EvolveBCartesian / EvolveECartesian with T_Algo = CartesianNodalAlgorithm (EvolveB.cpp:164-186, EvolveE.cpp:179-216)
void evolve_b_nodal(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double dinv[3]) {
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]);
    const NodalD D{dinv[0], dinv[1], dinv[2]};
#pragma omp parallel for
    for (int k = vlo(B[0], 2); k < vhi(B[0], 2); ++k)
        for (int j = vlo(B[0], 1); j < vhi(B[0], 1); ++j)
            for (int i = vlo(B[0], 0); i < vhi(B[0], 0); ++i) {
                Bx(i, j, k) += dt * D.Dz(Ey, i, j, k) - dt * D.Dy(Ez, i, j, k);
                By(i, j, k) += dt * D.Dx(Ez, i, j, k) - dt * D.Dz(Ex, i, j, k);
                Bz(i, j, k) += dt * D.Dy(Ex, i, j, k) - dt * D.Dx(Ey, i, j, k);
            }
}

void evolve_e_nodal(const wxa_field_view E[3], const wxa_field_view B[3], const wxa_field_view J[3], double dt,
                    const double dinv[3]) {
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]), jx(J[0]), jy(J[1]), jz(J[2]);
    const NodalD D{dinv[0], dinv[1], dinv[2]};
    constexpr double c2 = PhysConst::c * PhysConst::c;
#pragma omp parallel for
    for (int k = vlo(E[0], 2); k < vhi(E[0], 2); ++k)
        for (int j = vlo(E[0], 1); j < vhi(E[0], 1); ++j)
            for (int i = vlo(E[0], 0); i < vhi(E[0], 0); ++i) {
                Ex(i, j, k) += c2 * dt * (-D.Dz(By, i, j, k) + D.Dy(Bz, i, j, k) - PhysConst::mu0 * jx(i, j, k));
                Ey(i, j, k) += c2 * dt * (-D.Dx(Bz, i, j, k) + D.Dz(Bx, i, j, k) - PhysConst::mu0 * jy(i, j, k));
                Ez(i, j, k) += c2 * dt * (-D.Dy(Bx, i, j, k) + D.Dx(By, i, j, k) - PhysConst::mu0 * jz(i, j, k));
            }
}

}  // namespace

extern "C" {

const char* orc_version(void) { return "warpx_amd oracle (CPU restatement), fp64"; }
int orc_num_threads(void) { return max_threads(); }
// number of OpenMP threads of the following calls (bench.py's serial CPU figure: BASELINE.json config 1 is "CPU serial")
int orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
    return max_threads();
}

// Source/FieldSolver/FiniteDifferenceSolver/EvolveB.cpp:122-215 with
// CartesianYeeAlgorithm::UpwardD{x,y,z} (CartesianYeeAlgorithm.H:69-101,125-167,191-225)
static int evolve_b_clipped(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double dinv[3],
                            const int32_t* clo, const int32_t* chi) {
    if (all_nodal(E) && all_nodal(B)) { if (clo) return -3; evolve_b_nodal(E, B, dt, dinv); return 0; }
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]);
    const double idx = dinv[0], idy = dinv[1], idz = dinv[2];
#pragma omp parallel
    {
#pragma omp for nowait
        for (int k = clo_(B[0], 2, clo); k < chi_(B[0], 2, chi); ++k)
            for (int j = clo_(B[0], 1, clo); j < chi_(B[0], 1, chi); ++j)
                for (int i = clo_(B[0], 0, clo); i < chi_(B[0], 0, chi); ++i)
                    Bx(i, j, k) += dt * (idz * (Ey(i, j, k + 1) - Ey(i, j, k))) -
                                   dt * (idy * (Ez(i, j + 1, k) - Ez(i, j, k)));
#pragma omp for nowait
        for (int k = clo_(B[1], 2, clo); k < chi_(B[1], 2, chi); ++k)
            for (int j = clo_(B[1], 1, clo); j < chi_(B[1], 1, chi); ++j)
                for (int i = clo_(B[1], 0, clo); i < chi_(B[1], 0, chi); ++i)
                    By(i, j, k) += dt * (idx * (Ez(i + 1, j, k) - Ez(i, j, k))) -
                                   dt * (idz * (Ex(i, j, k + 1) - Ex(i, j, k)));
#pragma omp for nowait
        for (int k = clo_(B[2], 2, clo); k < chi_(B[2], 2, chi); ++k)
            for (int j = clo_(B[2], 1, clo); j < chi_(B[2], 1, chi); ++j)
                for (int i = clo_(B[2], 0, clo); i < chi_(B[2], 0, chi); ++i)
                    Bz(i, j, k) += dt * (idy * (Ex(i, j + 1, k) - Ex(i, j, k))) -
                                   dt * (idx * (Ey(i + 1, j, k) - Ey(i, j, k)));
    }
    return 0;
}

// The first guard layer of B next to the faces of the directions with grow[d] != 0, updated like the valid
// points from the guard points already present (see include/warpx_amd.h): components cell-centred along d,
// indices lo - 1 and lo + ncell, the component's valid range in the other two directions.
int orc_evolve_b_guard_layer(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double dinv[3],
                             const int32_t grow[3], void*) {
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]);
    const double idx = dinv[0], idy = dinv[1], idz = dinv[2];
    for (int d = 0; d < 3; ++d) {
        if (!grow[d]) continue;
        for (int side = 0; side < 2; ++side)
            for (int c = 0; c < 3; ++c) {
                if (B[c].stag[d]) continue;
                int lo[3], hi[3];
                for (int e = 0; e < 3; ++e) { lo[e] = vlo(B[c], e); hi[e] = vhi(B[c], e); }
                const int layer = side == 0 ? lo[d] - 1 : hi[d];
                lo[d] = layer; hi[d] = layer + 1;
                for (int k = lo[2]; k < hi[2]; ++k)
                    for (int j = lo[1]; j < hi[1]; ++j)
                        for (int i = lo[0]; i < hi[0]; ++i) {
                            if (c == 0)
                                Bx(i, j, k) += dt * (idz * (Ey(i, j, k + 1) - Ey(i, j, k))) -
                                               dt * (idy * (Ez(i, j + 1, k) - Ez(i, j, k)));
                            else if (c == 1)
                                By(i, j, k) += dt * (idx * (Ez(i + 1, j, k) - Ez(i, j, k))) -
                                               dt * (idz * (Ex(i, j, k + 1) - Ex(i, j, k)));
                            else
                                Bz(i, j, k) += dt * (idy * (Ex(i, j + 1, k) - Ex(i, j, k))) -
                                               dt * (idx * (Ey(i + 1, j, k) - Ey(i, j, k)));
                        }
            }
    }
    return 0;
}

// Source/FieldSolver/FiniteDifferenceSolver/EvolveE.cpp:120-250 (no EB, no F term)
static int evolve_e_clipped(const wxa_field_view E[3], const wxa_field_view B[3], const wxa_field_view J[3],
                            double dt, const double dinv[3], const int32_t* clo, const int32_t* chi) {
    if (all_nodal(E) && all_nodal(B) && all_nodal(J)) { if (clo) return -3; evolve_e_nodal(E, B, J, dt, dinv); return 0; }
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]), jx(J[0]), jy(J[1]), jz(J[2]);
    const double idx = dinv[0], idy = dinv[1], idz = dinv[2];
    constexpr double c2 = PhysConst::c * PhysConst::c;
#pragma omp parallel
    {
#pragma omp for nowait
        for (int k = clo_(E[0], 2, clo); k < chi_(E[0], 2, chi); ++k)
            for (int j = clo_(E[0], 1, clo); j < chi_(E[0], 1, chi); ++j)
                for (int i = clo_(E[0], 0, clo); i < chi_(E[0], 0, chi); ++i)
                    Ex(i, j, k) += c2 * dt *
                                   (-(idz * (By(i, j, k) - By(i, j, k - 1))) +
                                    (idy * (Bz(i, j, k) - Bz(i, j - 1, k))) - PhysConst::mu0 * jx(i, j, k));
#pragma omp for nowait
        for (int k = clo_(E[1], 2, clo); k < chi_(E[1], 2, chi); ++k)
            for (int j = clo_(E[1], 1, clo); j < chi_(E[1], 1, chi); ++j)
                for (int i = clo_(E[1], 0, clo); i < chi_(E[1], 0, chi); ++i)
                    Ey(i, j, k) += c2 * dt *
                                   (-(idx * (Bz(i, j, k) - Bz(i - 1, j, k))) +
                                    (idz * (Bx(i, j, k) - Bx(i, j, k - 1))) - PhysConst::mu0 * jy(i, j, k));
#pragma omp for nowait
        for (int k = clo_(E[2], 2, clo); k < chi_(E[2], 2, chi); ++k)
            for (int j = clo_(E[2], 1, clo); j < chi_(E[2], 1, chi); ++j)
                for (int i = clo_(E[2], 0, clo); i < chi_(E[2], 0, chi); ++i)
                    Ez(i, j, k) += c2 * dt *
                                   (-(idy * (Bx(i, j, k) - Bx(i, j - 1, k))) +
                                    (idx * (By(i, j, k) - By(i - 1, j, k))) - PhysConst::mu0 * jz(i, j, k));
    }
    return 0;
}


// ---- algo.maxwell_solver = ckc ---------------------------------------------------------------------------------
// CartesianCKCAlgorithm::InitializeStencilCoefficients, 3-D branch
// (Source/FieldSolver/FiniteDifferenceSolver/FiniteDifferenceAlgorithms/CartesianCKCAlgorithm.H:36-101)
void orc_ckc_stencil_coefficients(const double cell_size[3], double cx[5], double cy[5], double cz[5]) {
    const double inv_dx = 1. / cell_size[0], inv_dy = 1. / cell_size[1], inv_dz = 1. / cell_size[2];
    const double delta = std::max({inv_dx, inv_dy, inv_dz});
    const double rx = (inv_dx / delta) * (inv_dx / delta);
    const double ry = (inv_dy / delta) * (inv_dy / delta);
    const double rz = (inv_dz / delta) * (inv_dz / delta);
    const double beta = 0.125 * (1. - rx * ry * rz / (ry * rz + rz * rx + rx * ry));
    const double betaxy = ry * beta * inv_dx, betaxz = rz * beta * inv_dx;
    const double betayx = rx * beta * inv_dy, betayz = rz * beta * inv_dy;
    const double betazx = rx * beta * inv_dz, betazy = ry * beta * inv_dz;
    const double inv_r_fac = (1. / (ry * rz + rz * rx + rx * ry));
    const double gammax = ry * rz * (0.0625 - 0.125 * ry * rz * inv_r_fac);
    const double gammay = rx * rz * (0.0625 - 0.125 * rx * rz * inv_r_fac);
    const double gammaz = rx * ry * (0.0625 - 0.125 * rx * ry * inv_r_fac);
    const double alphax = (1. - 2. * ry * beta - 2. * rz * beta - 4. * gammax) * inv_dx;
    const double alphay = (1. - 2. * rx * beta - 2. * rz * beta - 4. * gammay) * inv_dy;
    const double alphaz = (1. - 2. * rx * beta - 2. * ry * beta - 4. * gammaz) * inv_dz;
    cx[0] = inv_dx; cx[1] = alphax; cx[2] = betaxy; cx[3] = betaxz; cx[4] = gammax * inv_dx;
    cy[0] = inv_dy; cy[1] = alphay; cy[2] = betayz; cy[3] = betayx; cy[4] = gammay * inv_dy;
    cz[0] = inv_dz; cz[1] = alphaz; cz[2] = betazx; cz[3] = betazy; cz[4] = gammaz * inv_dz;
}
// CartesianCKCAlgorithm::ComputeMaxDt (:107-120)
double orc_ckc_max_dt(const double cell_size[3]) {
    return std::min(cell_size[0], std::min(cell_size[1], cell_size[2])) / PhysConst::c;
}
namespace ckc {
// UpwardDx / UpwardDy / UpwardDz (:129-160, :183-214, :237-272), term by term in the reference's order
inline double up_x(const Arr& F, const double* c, int i, int j, int k) {
    const double alphax = c[1], betaxy = c[2], betaxz = c[3], gammax = c[4];
    return alphax * (F(i + 1, j, k) - F(i, j, k))
         + betaxy * (F(i + 1, j + 1, k) - F(i, j + 1, k)
                  +  F(i + 1, j - 1, k) - F(i, j - 1, k))
         + betaxz * (F(i + 1, j, k + 1) - F(i, j, k + 1)
                  +  F(i + 1, j, k - 1) - F(i, j, k - 1))
         + gammax * (F(i + 1, j + 1, k + 1) - F(i, j + 1, k + 1)
                  +  F(i + 1, j - 1, k + 1) - F(i, j - 1, k + 1)
                  +  F(i + 1, j + 1, k - 1) - F(i, j + 1, k - 1)
                  +  F(i + 1, j - 1, k - 1) - F(i, j - 1, k - 1));
}
inline double up_y(const Arr& F, const double* c, int i, int j, int k) {
    const double alphay = c[1], betayz = c[2], betayx = c[3], gammay = c[4];
    return alphay * (F(i, j + 1, k) - F(i, j, k))
         + betayx * (F(i + 1, j + 1, k) - F(i + 1, j, k)
                  +  F(i - 1, j + 1, k) - F(i - 1, j, k))
         + betayz * (F(i, j + 1, k + 1) - F(i, j, k + 1)
                  +  F(i, j + 1, k - 1) - F(i, j, k - 1))
         + gammay * (F(i + 1, j + 1, k + 1) - F(i + 1, j, k + 1)
                  +  F(i - 1, j + 1, k + 1) - F(i - 1, j, k + 1)
                  +  F(i + 1, j + 1, k - 1) - F(i + 1, j, k - 1)
                  +  F(i - 1, j + 1, k - 1) - F(i - 1, j, k - 1));
}
inline double up_z(const Arr& F, const double* c, int i, int j, int k) {
    const double alphaz = c[1], betazx = c[2], betazy = c[3], gammaz = c[4];
    return alphaz * (F(i, j, k + 1) - F(i, j, k))
         + betazx * (F(i + 1, j, k + 1) - F(i + 1, j, k)
                  +  F(i - 1, j, k + 1) - F(i - 1, j, k))
         + betazy * (F(i, j + 1, k + 1) - F(i, j + 1, k)
                  +  F(i, j - 1, k + 1) - F(i, j - 1, k))
         + gammaz * (F(i + 1, j + 1, k + 1) - F(i + 1, j + 1, k)
                  +  F(i - 1, j + 1, k + 1) - F(i - 1, j + 1, k)
                  +  F(i + 1, j - 1, k + 1) - F(i + 1, j - 1, k)
                  +  F(i - 1, j - 1, k + 1) - F(i - 1, j - 1, k));
}
}  // namespace ckc
// EvolveBCartesian<CartesianCKCAlgorithm> (EvolveB.cpp:164-186 with T_Algo = CKC)
int orc_evolve_b_ckc(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double cx[5],
                     const double cy[5], const double cz[5], void*) {
    const Arr Ex(E[0]), Ey(E[1]), Ez(E[2]), Bx(B[0]), By(B[1]), Bz(B[2]);
#pragma omp parallel
    {
#pragma omp for nowait
        for (int k = clo_(B[0], 2, nullptr); k < chi_(B[0], 2, nullptr); ++k)
            for (int j = clo_(B[0], 1, nullptr); j < chi_(B[0], 1, nullptr); ++j)
                for (int i = clo_(B[0], 0, nullptr); i < chi_(B[0], 0, nullptr); ++i)
                    Bx(i, j, k) += dt * ckc::up_z(Ey, cz, i, j, k) - dt * ckc::up_y(Ez, cy, i, j, k);
#pragma omp for nowait
        for (int k = clo_(B[1], 2, nullptr); k < chi_(B[1], 2, nullptr); ++k)
            for (int j = clo_(B[1], 1, nullptr); j < chi_(B[1], 1, nullptr); ++j)
                for (int i = clo_(B[1], 0, nullptr); i < chi_(B[1], 0, nullptr); ++i)
                    By(i, j, k) += dt * ckc::up_x(Ez, cx, i, j, k) - dt * ckc::up_z(Ex, cz, i, j, k);
#pragma omp for nowait
        for (int k = clo_(B[2], 2, nullptr); k < chi_(B[2], 2, nullptr); ++k)
            for (int j = clo_(B[2], 1, nullptr); j < chi_(B[2], 1, nullptr); ++j)
                for (int i = clo_(B[2], 0, nullptr); i < chi_(B[2], 0, nullptr); ++i)
                    Bz(i, j, k) += dt * ckc::up_y(Ex, cy, i, j, k) - dt * ckc::up_x(Ey, cx, i, j, k);
    }
    return 0;
}

int orc_evolve_b(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double dinv[3], void*) {
    return evolve_b_clipped(E, B, dt, dinv, nullptr, nullptr);
}
int orc_evolve_e(const wxa_field_view E[3], const wxa_field_view B[3], const wxa_field_view J[3], double dt,
                 const double dinv[3], void*) {
    return evolve_e_clipped(E, B, J, dt, dinv, nullptr, nullptr);
}
// the shell / interior pieces of an overlapped halo exchange: only the points inside the index box [lo, hi)
int orc_evolve_b_box(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double dinv[3],
                     const int32_t lo[3], const int32_t hi[3], void*) {
    return evolve_b_clipped(E, B, dt, dinv, lo, hi);
}
int orc_evolve_e_box(const wxa_field_view E[3], const wxa_field_view B[3], const wxa_field_view J[3], double dt,
                     const double dinv[3], const int32_t lo[3], const int32_t hi[3], void*) {
    return evolve_e_clipped(E, B, J, dt, dinv, lo, hi);
}

}  // extern "C"

namespace {

// GetExternalEBField::operator(), the RepeatedPlasmaLens branch and the transformation to the boosted frame
// (Source/Particles/Gather/GetExternalFields.H:137-189); m_uz_boost from GetExternalFields.cpp:28
inline void getExternalEB_lens(const wxa_repeated_plasma_lens& L, double time, double x, double y, double z, double uxp,
                               double uyp, double uzp, double& field_Ex, double& field_Ey, double& field_Bx,
                               double& field_By) {
    constexpr double c = 299792458.0;
    constexpr double inv_c2 = 1.0 / (c * c);
    const double uz_boost = std::sqrt(L.gamma_boost * L.gamma_boost - 1.0) * c;
    double Ex = 0.0, Ey = 0.0, Bx = 0.0, By = 0.0;
    const double gamma = std::sqrt(1.0 + (uxp * uxp + uyp * uyp + uzp * uzp) * inv_c2);
    const double vzp = uzp / gamma;
    double zl = z;
    double zr = z + vzp * L.dt;
    if (L.gamma_boost > 1.0) {
        zl = L.gamma_boost * zl + uz_boost * time;
        zr = L.gamma_boost * zr + uz_boost * (time + L.dt);
    }
    if (zl > 0) {
        const int i_lens = static_cast<int>(std::floor(zl / L.period));
        if (i_lens < L.n_lenses) {
            const double lens_start = L.starts[i_lens] + i_lens * L.period;
            const double lens_end = lens_start + L.lengths[i_lens];
            const double zl_bounded = std::min(std::max(zl, lens_start), lens_end);
            const double zr_bounded = std::min(std::max(zr, lens_start), lens_end);
            const double frac = ((zr - zl) == 0.0 ? 1.0 : (zr_bounded - zl_bounded) / (zr - zl));
            Ex += x * frac * L.strengths_E[i_lens];
            Ey += y * frac * L.strengths_E[i_lens];
            Bx += +y * frac * L.strengths_B[i_lens];
            By += -x * frac * L.strengths_B[i_lens];
        }
    }
    if (L.gamma_boost > 1.0) {
        const double Ex_boost = L.gamma_boost * Ex - uz_boost * By;
        const double Ey_boost = L.gamma_boost * Ey + uz_boost * Bx;
        const double Bx_boost = L.gamma_boost * Bx + uz_boost * Ey * inv_c2;
        const double By_boost = L.gamma_boost * By - uz_boost * Ex * inv_c2;
        Ex = Ex_boost; Ey = Ey_boost; Bx = Bx_boost; By = By_boost;
    }
    field_Ex += Ex; field_Ey += Ey; field_Bx += Bx; field_By += By;
}

template <int O, int G, bool MOVE>
void gather_push_impl(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                      const wxa_grid_geom* g, double q, double m, double dt, int pusher, const double* ext,
                      const wxa_repeated_plasma_lens* lens, double time) {
    const Arr ex(E[0]), ey(E[1]), ez(E[2]), bx(B[0]), by(B[1]), bz(B[2]);
    // particles.E/B_external_particle (constant): the sums of the gather start from them
    // (PhysicalParticleContainer.cpp:2589-2596,2705-2710)
    const double e0 = ext ? ext[0] : 0., e1 = ext ? ext[1] : 0., e2 = ext ? ext[2] : 0.;
    const double b0 = ext ? ext[3] : 0., b1 = ext ? ext[4] : 0., b2 = ext ? ext[5] : 0.;
#pragma omp parallel for schedule(static)
    for (int64_t ip = 0; ip < p->np; ++ip) {
        double xp = p->x[ip], yp = p->y[ip], zp = p->z[ip];
        double Exp = e0, Eyp = e1, Ezp = e2, Bxp = b0, Byp = b1, Bzp = b2;
        doGatherShapeN<O, G>(xp, yp, zp, Exp, Eyp, Ezp, Bxp, Byp, Bzp, ex, ey, ez, bx, by, bz,
                             E[0].stag, E[1].stag, E[2].stag, B[0].stag, B[1].stag, B[2].stag,
                             g->dinv, g->xyzmin, g->lo);
        if (lens && lens->n_lenses > 0)
            getExternalEB_lens(*lens, time, xp, yp, zp, p->ux[ip], p->uy[ip], p->uz[ip], Exp, Eyp, Bxp, Byp);
        doParticleMomentumPush(p->ux[ip], p->uy[ip], p->uz[ip], Exp, Eyp, Ezp, Bxp, Byp, Bzp, m, q,
                               pusher, dt);
        if (MOVE) {
            UpdatePosition(xp, yp, zp, p->ux[ip], p->uy[ip], p->uz[ip], dt);
            p->x[ip] = xp; p->y[ip] = yp; p->z[ip] = zp;
        }
    }
}

template <bool MOVE>
int gather_push_dispatch(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                         const wxa_grid_geom* g, double q, double m, double dt, int order, int galerkin,
                         int pusher, const double* ext = nullptr, const wxa_repeated_plasma_lens* lens = nullptr,
                         double time = 0.0) {
    // Source/Particles/Gather/FieldGather.H:1590-1664 (runtime dispatch on nox, galerkin)
    if (galerkin) {
        if (order == 1) gather_push_impl<1, 1, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 2) gather_push_impl<2, 1, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 3) gather_push_impl<3, 1, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 4) gather_push_impl<4, 1, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else return -1;
    } else {
        if (order == 1) gather_push_impl<1, 0, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 2) gather_push_impl<2, 0, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 3) gather_push_impl<3, 0, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else if (order == 4) gather_push_impl<4, 0, MOVE>(p, E, B, g, q, m, dt, pusher, ext, lens, time);
        else return -1;
    }
    return 0;
}

// Thread-private J scratch + locked accumulate, the reference's CPU tiling path
// (Source/Particles/WarpXParticleContainer.cpp:455-470,819-826).
struct PrivJ {
    std::vector<double> buf[3];
    wxa_field_view v[3];
};

template <int O>
void deposit_range(const wxa_particle_view* p, int64_t b, int64_t e, const wxa_field_view J[3],
                   const wxa_grid_geom* g, double q, double dt, double rel, int algo) {
    const Arr jx(J[0]), jy(J[1]), jz(J[2]);
    for (int64_t ip = b; ip < e; ++ip) {
        if (algo == WXA_DEPOSIT_ESIRKEPOV)
            doEsirkepovDepositionShapeN_one<O>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], p->ux[ip], p->uy[ip],
                                               p->uz[ip], jx, jy, jz, dt, rel, g->dinv, g->xyzmin, g->lo, q);
        else
            doDepositionShapeN_one<O>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], p->ux[ip], p->uy[ip],
                                      p->uz[ip], jx, jy, jz, J[0].stag, J[1].stag, J[2].stag, rel, g->dinv,
                                      g->xyzmin, g->lo, q);
    }
}

template <int O>
void deposit_impl(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* g, double q,
                  double dt, double rel, int algo) {
    const int nt = max_threads();
    if (nt <= 1 || p->np < 4096) {
        deposit_range<O>(p, 0, p->np, J, g, q, dt, rel, algo);
        return;
    }
    // Thread-private J scratch + accumulate (WarpXParticleContainer.cpp:455-470,819-826).  Each
    // thread takes a contiguous particle range; its scratch only spans the k-planes that range can
    // touch (the range's z extent + the stencil reach), so memory stays bounded on many-core hosts.
    // The scratch arrays are added to J plane by plane in thread order (not with atomics in arrival
    // order): for a given number of threads the result is reproducible bit for bit, which the tests
    // that compare two schedules by checksum rely on.
    const int reach = O + 3;
    std::vector<PrivJ> priv((size_t)nt);
#pragma omp parallel
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        const int64_t b = p->np * t / T, e = p->np * (t + 1) / T;
        PrivJ& pj = priv[(size_t)t];
        for (int c = 0; c < 3; ++c) { pj.v[c] = J[c]; pj.v[c].n[2] = 0; }
        if (e > b) {
            double zmin = p->z[b], zmax = p->z[b];
            for (int64_t i = b; i < e; ++i) { zmin = std::min(zmin, p->z[i]); zmax = std::max(zmax, p->z[i]); }
            int k0 = g->lo[2] + (int)std::floor((zmin - g->xyzmin[2]) * g->dinv[2]) - reach;
            int k1 = g->lo[2] + (int)std::floor((zmax - g->xyzmin[2]) * g->dinv[2]) + reach + 1;
            for (int c = 0; c < 3; ++c) {
                const int lo = std::max(k0, J[c].lo[2]), hi = std::min(k1, J[c].lo[2] + J[c].n[2]);
                pj.v[c].lo[2] = lo;
                pj.v[c].n[2] = std::max(hi - lo, 0);
                pj.buf[c].assign((size_t)J[c].kstride * pj.v[c].n[2], 0.0);
                pj.v[c].p = pj.buf[c].data();
            }
            deposit_range<O>(p, b, e, pj.v, g, q, dt, rel, algo);
        }
#pragma omp barrier
        for (int c = 0; c < 3; ++c) {
            const size_t plane = (size_t)J[c].kstride;
#pragma omp for schedule(static)
            for (int k = J[c].lo[2]; k < J[c].lo[2] + J[c].n[2]; ++k) {
                double* dst = J[c].p + (size_t)(k - J[c].lo[2]) * plane;
                for (int tt = 0; tt < T; ++tt) {
                    const PrivJ& src_j = priv[(size_t)tt];
                    if (k < src_j.v[c].lo[2] || k >= src_j.v[c].lo[2] + src_j.v[c].n[2]) continue;
                    const double* src = src_j.buf[c].data() + (size_t)(k - src_j.v[c].lo[2]) * plane;
                    for (size_t i = 0; i < plane; ++i) dst[i] += src[i];
                }
            }
        }
    }
}

}  // namespace

extern "C" {

// Source/Particles/PhysicalParticleContainer.cpp:2549-2786 (PushPX), kernel :2687-2785
int orc_gather_push(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                    const wxa_grid_geom* g, double q, double m, double dt, int order, int galerkin,
                    int pusher, void*) {
    return gather_push_dispatch<true>(p, E, B, g, q, m, dt, order, galerkin, pusher);
}

// PushPX (move != 0) / PushP with the container's constant external fields ext = {Ex,Ey,Ez,Bx,By,Bz} (or null)
int orc_gather_push_ext(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                        const wxa_grid_geom* g, double q, double m, double dt, int order, int galerkin, int pusher,
                        int move, const double* ext) {
    return move ? gather_push_dispatch<true>(p, E, B, g, q, m, dt, order, galerkin, pusher, ext)
                : gather_push_dispatch<false>(p, E, B, g, q, m, dt, order, galerkin, pusher, ext);
}

// the same with a repeated plasma lens evaluated at `time` (wxa_repeated_plasma_lens of include/warpx_amd.h, host arrays)
int orc_gather_push_lens(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                         const wxa_grid_geom* g, double q, double m, double dt, int order, int galerkin, int pusher,
                         int move, const double* ext, const wxa_repeated_plasma_lens* lens, double time) {
    return move ? gather_push_dispatch<true>(p, E, B, g, q, m, dt, order, galerkin, pusher, ext, lens, time)
                : gather_push_dispatch<false>(p, E, B, g, q, m, dt, order, galerkin, pusher, ext, lens, time);
}

// Source/Particles/PhysicalParticleContainer.cpp:2368-2516 (PushP), kernel :2454-2511
int orc_push_p(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
               const wxa_grid_geom* g, double q, double m, double dt, int order, int galerkin, int pusher,
               void*) {
    return gather_push_dispatch<false>(p, E, B, g, q, m, dt, order, galerkin, pusher);
}

// Source/Particles/WarpXParticleContainer.cpp:352-827 (DepositCurrent dispatch :481-816)
int orc_deposit_current(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* g,
                        double q, double dt, double relative_time, int order, int algo, void*, void*) {
    if (order == 1) deposit_impl<1>(p, J, g, q, dt, relative_time, algo);
    else if (order == 2) deposit_impl<2>(p, J, g, q, dt, relative_time, algo);
    else if (order == 3) deposit_impl<3>(p, J, g, q, dt, relative_time, algo);
    else if (order == 4) deposit_impl<4>(p, J, g, q, dt, relative_time, algo);
    else return -1;
    return 0;
}

int orc_deposit_charge(const wxa_particle_view* p, const wxa_field_view* rho, const wxa_grid_geom* g,
                       double q, int order, void*) {
    const Arr r(*rho);
    for (int64_t ip = 0; ip < p->np; ++ip) {
        if (order == 1) doChargeDepositionShapeN_one<1>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], r, rho->stag, g->dinv, g->xyzmin, g->lo, q);
        else if (order == 2) doChargeDepositionShapeN_one<2>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], r, rho->stag, g->dinv, g->xyzmin, g->lo, q);
        else if (order == 3) doChargeDepositionShapeN_one<3>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], r, rho->stag, g->dinv, g->xyzmin, g->lo, q);
        else if (order == 4) doChargeDepositionShapeN_one<4>(p->x[ip], p->y[ip], p->z[ip], p->w[ip], r, rho->stag, g->dinv, g->xyzmin, g->lo, q);
        else return -1;
    }
    return 0;
}

// Periodic wrap applied by amrex ParticleContainer::Redistribute (AMReX not in tree;
// restated contract: positions end up in [plo, phi]).
int orc_enforce_periodic(const wxa_particle_view* p, const double plo[3], const double phi[3],
                         const int periodic[3], void*) {
    double* pos[3] = {p->x, p->y, p->z};
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d]) continue;
        const double L = phi[d] - plo[d];
        double* a = pos[d];
#pragma omp parallel for
        for (int64_t ip = 0; ip < p->np; ++ip) {
            double v = a[ip];
            if (v >= phi[d]) {
                v -= L;
                if (v < plo[d]) v = plo[d];
            } else if (v < plo[d]) {
                v += L;
                if (v >= phi[d]) v = std::nextafter(phi[d], plo[d]);
            }
            a[ip] = v;
        }
    }
    return 0;
}

// Source/Filter/BilinearFilter.cpp:26-94 (npass = 1 -> stencil {0.5*0.5, 0.25}) and
// Source/Filter/Filter.cpp:92-133 (8 mirrored taps, zero padding outside the array).
int orc_filter_bilinear(const wxa_field_view* src, const wxa_field_view* dst, void*) {
    const Arr s(*src), d(*dst);
    // compute_stencil(npass=1): old_s = {1,0} -> new_s[0] = 0.5, new_s[1] = 0.25; old_s[0] *= 0.5
    const double st[2] = {0.25, 0.25};
    const int lo0 = src->lo[0], lo1 = src->lo[1], lo2 = src->lo[2];
    const int hi0 = lo0 + src->n[0], hi1 = lo1 + src->n[1], hi2 = lo2 + src->n[2];
    auto zp = [&](int i, int j, int k) -> double {
        return (i >= lo0 && i < hi0 && j >= lo1 && j < hi1 && k >= lo2 && k < hi2) ? s(i, j, k) : 0.0;
    };
#pragma omp parallel for collapse(2)
    for (int k = lo2; k < hi2; ++k)
        for (int j = lo1; j < hi1; ++j)
            for (int i = lo0; i < hi0; ++i) {
                double acc = 0.0;
                for (int i2 = 0; i2 < 2; ++i2)
                    for (int i1 = 0; i1 < 2; ++i1)
                        for (int i0 = 0; i0 < 2; ++i0) {
                            const double sss = st[i0] * st[i1] * st[i2];
                            acc += sss * (zp(i - i0, j - i1, k - i2) + zp(i + i0, j - i1, k - i2) +
                                          zp(i - i0, j + i1, k - i2) + zp(i + i0, j + i1, k - i2) +
                                          zp(i - i0, j - i1, k + i2) + zp(i + i0, j - i1, k + i2) +
                                          zp(i - i0, j + i1, k + i2) + zp(i + i0, j + i1, k + i2));
                        }
                d(i, j, k) = acc;
            }
    return 0;
}

// Filter::DoFilter (Source/Filter/Filter.cpp:92-133) for any stencil lengths (the NCI corrector's filter has 1, 1, 5)
int orc_filter_stencil(const wxa_field_view* src, const wxa_field_view* dst, const double* s0, int32_t n0,
                       const double* s1, int32_t n1, const double* s2, int32_t n2, void*) {
    const Arr s(*src), d(*dst);
    const int lo0 = src->lo[0], lo1 = src->lo[1], lo2 = src->lo[2];
    const int hi0 = lo0 + src->n[0], hi1 = lo1 + src->n[1], hi2 = lo2 + src->n[2];
    auto zp = [&](int i, int j, int k) -> double {   // src_zeropad (:111-114)
        return (i >= lo0 && i < hi0 && j >= lo1 && j < hi1 && k >= lo2 && k < hi2) ? s(i, j, k) : 0.0;
    };
#pragma omp parallel for collapse(2)
    for (int k = lo2; k < hi2; ++k)
        for (int j = lo1; j < hi1; ++j)
            for (int i = lo0; i < hi0; ++i) {
                double acc = 0.0;
                for (int i2 = 0; i2 < n2; ++i2)
                    for (int i1 = 0; i1 < n1; ++i1)
                        for (int i0 = 0; i0 < n0; ++i0) {
                            const double sss = s0[i0] * s1[i1] * s2[i2];
                            acc += sss * (zp(i - i0, j - i1, k - i2) + zp(i + i0, j - i1, k - i2) +
                                          zp(i - i0, j + i1, k - i2) + zp(i + i0, j + i1, k - i2) +
                                          zp(i - i0, j - i1, k + i2) + zp(i + i0, j - i1, k + i2) +
                                          zp(i - i0, j + i1, k + i2) + zp(i + i0, j + i1, k + i2));
                        }
                d(i, j, k) = acc;
            }
    return 0;
}

// NCIGodfreyFilter::ComputeStencils (Source/Filter/NCIGodfreyFilter.cpp:45-154); the coefficient tables are the data
// file the product ships too (generated from Source/Utils/NCIGodfreyTables.H, pinned against it by
// tests/test_nci_cpu.py where the reference is on disk)
int orc_nci_godfrey_stencil(double cdtodz, int32_t nodal_gather, int32_t coeff_set, double stencil_z[5]) {
    using namespace orc_nci_godfrey;
    if (coeff_set != 0 && coeff_set != 1) return -1;
    int index = static_cast<int>(tab_length * cdtodz);                     // :57
    index = std::min(index, tab_length - 2);                               // :58
    index = std::max(index, 0);                                            // :59
    const double weight_right = cdtodz - double(index) / double(tab_length);   // :60
    const double* tab = tables + (size_t)((nodal_gather ? 2 : 0) + coeff_set) * tab_length * tab_width;
    double pre[4];
    for (int i = 0; i < tab_width; i++)                                    // :66-98
        pre[i] = (1. - weight_right) * tab[index * tab_width + i] + weight_right * tab[(index + 1) * tab_width + i];
    stencil_z[0] = (256 + 128 * pre[0] + 96 * pre[1] + 80 * pre[2] + 70 * pre[3]) / 256;   // :101-105
    stencil_z[1] = -(64 * pre[0] + 64 * pre[1] + 60 * pre[2] + 56 * pre[3]) / 256;
    stencil_z[2] = (16 * pre[1] + 24 * pre[2] + 28 * pre[3]) / 256;
    stencil_z[3] = -(4 * pre[2] + 8 * pre[3]) / 256;
    stencil_z[4] = (1 * pre[3]) / 256;
    stencil_z[0] /= 2.;                                                    // :126
    return 0;
}

// amrex FabArray::FillBoundary(ng, periodicity) on a single periodic brick
// (contract: SURVEY.md Appendix B; call sites Source/ablastr/utils/Communication.cpp:108-113).
// Direction by direction so that edges/corners are filled from already-filled lines.
int orc_fill_boundary_periodic(const wxa_field_view* f, const int ng[3], const int periodic[3], void*) {
    const Arr a(*f);
    int lo[3], hi[3];  // region already consistent: starts as the valid box
    for (int d = 0; d < 3; ++d) { lo[d] = vlo(*f, d); hi[d] = vhi(*f, d); }
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d] || ng[d] <= 0) continue;
        const int nc = ncell_of(*f, d);
        const int v0 = vlo(*f, d), v1 = vhi(*f, d);
        // the lines along d are independent of each other: all cores, one parallel region per direction and side (the
        // reference's OpenMP build runs FillBoundary on all threads too).  Threads split the SLOWEST of the other two
        // dimensions and keep whole rows: splitting i between threads would have them write into one another's cache lines
        const int dp = d == 2 ? 1 : 2, dq = d == 0 ? 1 : 0;
        // (a parallel region costs tens of microseconds on a 128-core host: small arrays -- the 16^3 .. 32^3 grids of the
        // parity tests, thousands of steps -- stay serial, with no OpenMP construct on their path at all)
        const bool big = (long)(hi[dp] - lo[dp]) * (hi[dq] - lo[dq]) * ng[d] >= 65536;
        for (int side = 0; side < 2; ++side) {
            auto plane = [&](const int u) {
                // guard layers in ascending depth (a layer deeper than the period reads a layer filled just before);
                // the contiguous index innermost where it is not the exchanged direction itself
                for (int gi = 1; gi <= ng[d]; ++gi) {
                    const int dsti = side == 0 ? v0 - gi : v1 - 1 + gi;
                    const int srci = side == 0 ? dsti + nc : dsti - nc;
                    for (int v = lo[dq]; v < hi[dq]; ++v) {
                        int t[3], s[3];
                        t[dq] = v; t[dp] = u; t[d] = dsti;
                        s[dq] = v; s[dp] = u; s[d] = srci;
                        a(t[0], t[1], t[2]) = a(s[0], s[1], s[2]);
                    }
                }
            };
            if (big) {
#pragma omp parallel for
                for (int u = lo[dp]; u < hi[dp]; ++u) plane(u);
            } else {
                for (int u = lo[dp]; u < hi[dp]; ++u) plane(u);
            }
        }
        lo[d] = v0 - ng[d]; hi[d] = v1 + ng[d];
    }
    return 0;
}

}  // extern "C"

// ---- PEC field boundary (Source/BoundaryConditions/WarpX_PEC.cpp) -------------------------
namespace {
// get_cell_count_to_boundary (:41-49): how many grid points `ijk` lies beyond the face; the high
// face sits between cells dom_hi and dom_hi+1, i.e. on node dom_hi+1 of a nodal direction.
inline int cells_beyond_face(const int32_t dom_lo[3], const int32_t dom_hi[3], const int ijk[3], const int nodal[3],
                             int idim, int iside) {
    return iside == 0 ? dom_lo[idim] - ijk[idim] : ijk[idim] - (dom_hi[idim] + nodal[idim]);
}

// SetEfieldOnPEC (:117-196) for IS_E, SetBfieldOnPEC (:256-331) otherwise: the two differ only in
// which components are "flagged" at a face -- the tangential ones for E, the normal one for B.
// A flagged component is zero on the face (if it lives on it) and odd across it; the others are even.
template <bool IS_E>
inline void set_field_on_pec(int icomp, const int32_t dom_lo[3], const int32_t dom_hi[3], const int ijk[3],
                             const Arr& f, const int nodal[3], const int32_t pec_lo[3], const int32_t pec_hi[3]) {
    int mirror[3] = {ijk[0], ijk[1], ijk[2]};
    bool on_face = false, guard = false;
    double sign = 1.0;
    for (int idim = 0; idim < 3; ++idim)
        for (int iside = 0; iside < 2; ++iside) {
            if (!(iside == 0 ? pec_lo[idim] : pec_hi[idim])) continue;
            const bool flagged = IS_E ? (icomp != idim) : (icomp == idim);
            const int ig = cells_beyond_face(dom_lo, dom_hi, ijk, nodal, idim, iside);
            if (ig == 0) {
                if (flagged && nodal[idim] == 1) on_face = true;
            } else if (ig > 0) {
                mirror[idim] = iside == 0 ? dom_lo[idim] + ig - (1 - nodal[idim]) : dom_hi[idim] + 1 - ig;
                guard = true;
                if (flagged) sign *= -1.0;
            }
        }
    if (on_face) f(ijk[0], ijk[1], ijk[2]) = 0.0;
    else if (guard) f(ijk[0], ijk[1], ijk[2]) = sign * f(mirror[0], mirror[1], mirror[2]);
}

// ApplyPECtoEfield (:457-538) / ApplyPECtoBfield (:540-626): every point of tilebox(ixType, ng)
template <bool IS_E>
int apply_pec(const wxa_field_view F[3], const int32_t dom_lo[3], const int32_t dom_hi[3], const int32_t pec_lo[3],
              const int32_t pec_hi[3], const int32_t ng[3]) {
    for (int c = 0; c < 3; ++c) {
        const Arr a(F[c]);
        const int nodal[3] = {F[c].stag[0], F[c].stag[1], F[c].stag[2]};
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d) {
            if (ng[d] > F[c].ng[d]) return -1;
            lo[d] = vlo(F[c], d) - ng[d];
            hi[d] = vhi(F[c], d) + ng[d];
        }
        for (int k = lo[2]; k < hi[2]; ++k)
            for (int j = lo[1]; j < hi[1]; ++j)
                for (int i = lo[0]; i < hi[0]; ++i) {
                    const int ijk[3] = {i, j, k};
                    set_field_on_pec<IS_E>(c, dom_lo, dom_hi, ijk, a, nodal, pec_lo, pec_hi);
                }
    }
    return 0;
}
}  // namespace

extern "C" {

// PEC::ApplyReflectiveBoundarytoJfield (:713-900) with SetRhoOrJfieldFromPEC (:354-420), for PEC field
// boundaries with absorbing particle boundaries (the default next to a PEC wall): the current
// deposited in the guard cells behind the wall is folded back onto its mirror cell with the sign of an
// image charge (psign = -1 for the components tangential to the wall, +1 for the normal one), a
// component living on the wall is zeroed there, and the guard cells then receive the image of the
// updated interior values (odd for tangential, even for normal components).
}  // extern "C"

namespace {
// SetRhoOrJfieldFromPEC (:354-420) over the valid box of one array; tangent[d]: odd image along d
// (psign -1), otherwise even (psign +1) -- absorbing particle boundaries
// transverse_guards: also fold the guard columns of the directions that have no PEC wall (see orc_apply_pec_rho)
void reflect_over_pec(const wxa_field_view& f, const bool tangent[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                      const int32_t pec_lo[3], const int32_t pec_hi[3], bool transverse_guards = false) {
    const Arr a(f);
    int mirrorfac[3][2];
    for (int d = 0; d < 3; ++d) {
        // the domain box made nodal: [dom_lo, dom_hi + 1]  (:729-733, :800-805; rho :640-676)
        mirrorfac[d][0] = 2 * dom_lo[d] - (1 - f.stag[d]);
        mirrorfac[d][1] = 2 * (dom_hi[d] + 1) - (1 - f.stag[d]);
    }
    auto in_fab = [&](const int v[3]) {
        for (int d = 0; d < 3; ++d)
            if (v[d] < f.lo[d] || v[d] >= f.lo[d] + f.n[d]) return false;
        return true;
    };
    int blo[3], bhi[3];
    for (int d = 0; d < 3; ++d) {
        const bool grow = transverse_guards && !pec_lo[d] && !pec_hi[d];
        blo[d] = grow ? f.lo[d] : vlo(f, d);
        bhi[d] = grow ? f.lo[d] + f.n[d] : vhi(f, d);
    }
    for (int k = blo[2]; k < bhi[2]; ++k)
        for (int j = blo[1]; j < bhi[1]; ++j)
            for (int i = blo[0]; i < bhi[0]; ++i) {
                const int ijk[3] = {i, j, k};
                for (int d = 0; d < 3; ++d)
                    for (int side = 0; side < 2; ++side) {
                        if (!(side == 0 ? pec_lo[d] : pec_hi[d])) continue;
                        int m[3] = {i, j, k};
                        m[d] = mirrorfac[d][side] - ijk[d];
                        if (m[d] == ijk[d]) a(i, j, k) = 0.0;
                        else if (in_fab(m)) a(i, j, k) += (tangent[d] ? -1.0 : 1.0) * a(m[0], m[1], m[2]);
                    }
                for (int d = 0; d < 3; ++d)
                    for (int side = 0; side < 2; ++side) {
                        if (!(side == 0 ? pec_lo[d] : pec_hi[d])) continue;
                        int m[3] = {i, j, k};
                        m[d] = mirrorfac[d][side] - ijk[d];
                        if (m[d] != ijk[d] && in_fab(m)) a(m[0], m[1], m[2]) = tangent[d] ? -a(i, j, k) : a(i, j, k);
                    }
            }
}
}  // namespace

extern "C" {

// PEC::ApplyReflectiveBoundarytoJfield (:713-900) with SetRhoOrJfieldFromPEC (:354-420), for PEC field
// boundaries with absorbing particle boundaries (the default next to a PEC wall): the current
// deposited in the guard cells behind the wall is folded back onto its mirror cell with the sign of an
// image charge (psign = -1 for the components tangential to the wall, +1 for the normal one), a
// component living on the wall is zeroed there, and the guard cells then receive the image of the
// updated interior values (odd for tangential, even for normal components).
int orc_apply_pec_j(const wxa_field_view J[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                    const int32_t pec_lo[3], const int32_t pec_hi[3], void*) {
    for (int c = 0; c < 3; ++c) {
        const bool tangent[3] = {c != 0, c != 1, c != 2};
        reflect_over_pec(J[c], tangent, dom_lo, dom_hi, pec_lo, pec_hi);
    }
    return 0;
}

// PEC::ApplyReflectiveBoundarytoRhofield (:628-711): rho is treated like a tangential component along
// every direction (:664-666).  The reference folds the valid points of each box only (:697), although
// it runs before the guard sum -- charge deposited in a box's transverse guard columns (towards a
// periodic image or a neighbour box) then misses its image, so its rho next to a wall depends on the
// box decomposition.  THIS entry point (the kernel-level one, mirrored by wxa_apply_pec_rho) folds the
// guard columns of the wall-free directions too, which is what a fold after the sum would give; the
// callers that want the reference's numbers -- the host layer's WarpX::ApplyRhofieldBoundary and the
// stepper below -- leave those columns as they are.  (Round 5: the difference is 1.3e-3 of the sum of
// |rho| in the reference's back-transformed golden run, whose plasma ends one cell from the periodic
// faces and streams through the lower wall.)
int orc_apply_pec_rho(const wxa_field_view* rho, const int32_t dom_lo[3], const int32_t dom_hi[3],
                      const int32_t pec_lo[3], const int32_t pec_hi[3], void*) {
    const bool tangent[3] = {true, true, true};
    reflect_over_pec(*rho, tangent, dom_lo, dom_hi, pec_lo, pec_hi, /*transverse_guards=*/true);
    return 0;
}

// WarpX::shiftMF (Source/Utils/WarpXMovingWindow.cpp:478-648), zero external field, forward window
int orc_shift_field_window(const wxa_field_view* f, double* tmp, int32_t dir, int32_t num_shift, const int periodic[3],
                           void*) {
    const wxa_field_view& v = *f;
    if (dir < 0 || dir > 2 || num_shift < 0 || num_shift > v.ng[dir]) return -1;
    if (num_shift == 0) return 0;
    const size_t count = (size_t)v.kstride * v.n[2];
    std::memcpy(tmp, v.p, sizeof(double) * count);   // MultiFab::Copy(tmpmf, mf, ..., ng)
    wxa_field_view tv = v;
    tv.p = tmp;
    // FillBoundary(tmpmf, ng_mw, periodicity): one guard cell, num_shift along the window direction
    int ng_mw[3] = {1, 1, 1};
    ng_mw[dir] = num_shift;
    for (int d = 0; d < 3; ++d) ng_mw[d] = std::min(ng_mw[d], (int)v.ng[d]);
    // WXA_WINDOW_KEEP_GUARDS: bricks along the window -- the guards beyond the high face hold the next brick's cells
    const bool keep_guards = periodic[dir] == WXA_WINDOW_KEEP_GUARDS;
    const int per[3] = {dir == 0 ? 0 : periodic[0], dir == 1 ? 0 : periodic[1], dir == 2 ? 0 : periodic[2]};
    orc_fill_boundary_periodic(&tv, ng_mw, per, nullptr);
    const Arr src(tv), dst(v);
    // the region the window moved into takes the external field (0): adjCellHi(domain, dir, ng) in the
    // field's index type, without the boundary node of a nodal direction, grown by ng transversally
    const int vhi_d = vhi(v, dir);   // exclusive end of the valid points along dir
    int zlo[3], zhi[3];
    for (int d = 0; d < 3; ++d) { zlo[d] = v.lo[d]; zhi[d] = v.lo[d] + v.n[d]; }
    zlo[dir] = vhi_d;                // first point beyond the domain (cell dom_hi+1 / node dom_hi+2)
    zhi[dir] = vhi_d + v.ng[dir];
    if (!keep_guards)
        for (int k = zlo[2]; k < zhi[2]; ++k)
            for (int j = zlo[1]; j < zhi[1]; ++j)
                for (int i = zlo[0]; i < zhi[0]; ++i) src(i, j, k) = 0.0;
    // dst(i) = src(i + shift) on the fab box shrunk by num_shift on the high side
    int dlo[3], dhi[3];
    for (int d = 0; d < 3; ++d) { dlo[d] = v.lo[d]; dhi[d] = v.lo[d] + v.n[d]; }
    dhi[dir] -= num_shift;
    int sh[3] = {0, 0, 0};
    sh[dir] = num_shift;
    for (int k = dlo[2]; k < dhi[2]; ++k)
        for (int j = dlo[1]; j < dhi[1]; ++j)
            for (int i = dlo[0]; i < dhi[0]; ++i) dst(i, j, k) = src(i + sh[0], j + sh[1], k + sh[2]);
    return 0;
}

// PhysicalParticleContainer::AddPlasma (:924-1333) for the cells [0, ncells) above `corner`, NUniformPerCell with a
// constant density, momentum u c (NULL = at rest): the counterpart of wxa_add_plasma, in lattice order
namespace {
// Philox4x32-10 + Box-Muller: the stream include/warpx_amd.h specifies for the injected gaussian momenta
void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
double uniform53(uint32_t hi, uint32_t lo) { return ((double)((((uint64_t)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
void normal3(uint64_t seed, int sx, int sy, int sz, double n[3]) {
    uint32_t a[4] = {(uint32_t)sx, (uint32_t)sy, (uint32_t)sz, 0u}, b[4] = {(uint32_t)sx, (uint32_t)sy, (uint32_t)sz, 1u};
    philox4x32_10(a, (uint32_t)seed, (uint32_t)(seed >> 32));
    philox4x32_10(b, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double r0 = std::sqrt(-2.0 * std::log(uniform53(a[0], a[1]))), t0 = 2.0 * M_PI * uniform53(a[2], a[3]);
    const double r1 = std::sqrt(-2.0 * std::log(uniform53(b[0], b[1]))), t1 = 2.0 * M_PI * uniform53(b[2], b[3]);
    n[0] = r0 * std::cos(t0); n[1] = r0 * std::sin(t0); n[2] = r1 * std::cos(t1);
}
}  // namespace

int orc_add_plasma(const wxa_particle_view* dst, const wxa_plasma_injector* inj, const double corner[3],
                   const int32_t ncells[3], const double dx[3], const double brick_lo[3], const double brick_hi[3],
                   const wxa_injected_momentum* mom, int64_t* n_added, void*, void*) {
    const bool thermal = mom && (mom->u_th[0] != 0.0 || mom->u_th[1] != 0.0 || mom->u_th[2] != 0.0);
    *n_added = 0;
    if (!(inj->density > 0)) return 0;
    const int nppc = inj->ppc[0] * inj->ppc[1] * inj->ppc[2];
    const double scale_fac = dx[0] * dx[1] * dx[2] / nppc;
    // applyBallisticCorrection (PhysicalParticleContainer.cpp:138-148) with the bulk momentum of the injector
    const double gamma_boost = inj->gamma_boost > 1.0 ? inj->gamma_boost : 1.0;
    const double beta_boost = gamma_boost > 1.0 ? std::sqrt(1.0 - 1.0 / std::pow(gamma_boost, 2.0)) : 0.0;
    const double ub[3] = {mom ? mom->u_mean[0] : 0.0, mom ? mom->u_mean[1] : 0.0, mom ? mom->u_mean[2] : 0.0};
    const double gamma_bulk = std::sqrt(1.0 + (ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2]));
    const double betaz_bulk = ub[2] / gamma_bulk;
    const double za = 1.0 - beta_boost * betaz_bulk, zb = PhysConst::c * inj->t * (betaz_bulk - beta_boost);
    auto ballistic = [&](double z) { return gamma_boost * (z * za - zb); };
    int64_t n = 0;
    for (int k = 0; k < ncells[2]; ++k)
        for (int j = 0; j < ncells[1]; ++j)
            for (int i = 0; i < ncells[0]; ++i) {
                const int iv[3] = {i, j, k};
                bool cell_ok = true;
                for (int d = 0; d < 3; ++d) {
                    double clo = corner[d] + (iv[d] + 0.0) * dx[d], chi = corner[d] + (iv[d] + 1.0) * dx[d];
                    if (d == 2) { clo = ballistic(clo); chi = ballistic(chi); }   // :1021-1022
                    const double mid = (clo + chi) / 2.;
                    const bool sample = (clo < inj->hi[d] && clo >= inj->lo[d]) || (mid < inj->hi[d] && mid >= inj->lo[d]) ||
                                        (chi < inj->hi[d] && chi >= inj->lo[d]);
                    cell_ok = cell_ok && !(clo > inj->hi[d] || chi < inj->lo[d]) && sample;
                }
                if (!cell_ok) continue;
                for (int ip = 0; ip < nppc; ++ip) {
                    const int ny = inj->ppc[1], nz = inj->ppc[2];
                    const int ix_part = ip / (ny * nz);
                    const int iz_part = (ip - ix_part * (ny * nz)) / ny;
                    const int iy_part = (ip - ix_part * (ny * nz)) - ny * iz_part;
                    const double r[3] = {(0.5 + ix_part) / inj->ppc[0], (0.5 + iy_part) / inj->ppc[1], (0.5 + iz_part) / inj->ppc[2]};
                    double pos[3];
                    bool ok = true;
                    for (int d = 0; d < 3; ++d) {
                        pos[d] = corner[d] + (iv[d] + r[d]) * dx[d];
                        const double lab = d == 2 ? ballistic(pos[d]) : pos[d];   // z0 / z0_lab of :1181, :1212
                        ok = ok && pos[d] > brick_lo[d] && pos[d] < brick_hi[d] && lab < inj->hi[d] && lab >= inj->lo[d];
                    }
                    if (!ok) continue;
                    if (n >= dst->np) return -4;
                    dst->x[n] = pos[0]; dst->y[n] = pos[1]; dst->z[n] = pos[2];
                    double u[3] = {ub[0], ub[1], ub[2]};
                    if (thermal) {
                        double nrm[3];
                        normal3(mom->seed, (int)std::floor((pos[0] - mom->origin[0]) / dx[0] * inj->ppc[0]),
                                (int)std::floor((pos[1] - mom->origin[1]) / dx[1] * inj->ppc[1]),
                                (int)std::floor((pos[2] - mom->origin[2]) / dx[2] * inj->ppc[2]), nrm);
                        for (int d = 0; d < 3; ++d) u[d] += mom->u_th[d] * nrm[d];
                    }
                    double dens = inj->density;
                    if (gamma_boost > 1.0) {   // :1232-1246 Lorentz transform of the lab-frame density and momentum
                        const double gamma_lab = std::sqrt(1.0 + (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
                        const double betaz_lab = u[2] / gamma_lab;
                        dens = gamma_boost * dens * (1.0 - beta_boost * betaz_lab);
                        u[2] = gamma_boost * (u[2] - beta_boost * gamma_lab);
                    }
                    dst->w[n] = dens * scale_fac;
                    dst->ux[n] = u[0] * PhysConst::c; dst->uy[n] = u[1] * PhysConst::c; dst->uz[n] = u[2] * PhysConst::c;
                    if (dst->idcpu) dst->idcpu[n] = 0;
                    ++n;
                }
            }
    *n_added = n;
    return 0;
}

// calculate_laser_plane_coordinates (LaserParticleContainer.cpp:795-847), GaussianLaserProfile::fill_amplitude
// (LaserProfileGaussian.cpp:104-161 with zeta = beta = phi2 = phi0 = 0, theta_stc = 0), update_laser_particle
// (:850-951); t is the lab-frame time also in a boosted frame (the caller converts, :574-579)
int orc_laser_push(const wxa_particle_view* p, const wxa_laser_push_params* c, double t, double dt, void*) {
    using cplx = std::complex<double>;
    const cplx I(0, 1);
    const double k0 = 2. * M_PI / c->wavelength;
    const double inv_tau2 = 1. / (c->duration * c->duration);
    const double oscillation_phase = k0 * PhysConst::c * (t - c->t_peak) + 0.0;
    const cplx diffract_factor = 1. + I * c->focal_distance * 2. / (k0 * c->waist * c->waist);
    const cplx inv_complex_waist_2 = 1. / (c->waist * c->waist * diffract_factor);
    const cplx stretch_factor = 1.;   // 1 + 4 (zeta + beta f / tau^2)(...) + 2 i (phi2 - ...) / tau^2 with zeta = beta = phi2 = 0
    const cplx t_prefactor = c->e_max * std::exp(I * oscillation_phase);
    const cplx prefactor = t_prefactor / diffract_factor;
    const double gamma_boost = c->gamma_boost > 1. ? c->gamma_boost : 1.;
    const double beta_boost = gamma_boost > 1. ? std::sqrt(1. - 1. / std::pow(gamma_boost, 2.)) : 0.;
    for (int64_t i = 0; i < p->np; ++i) {
        double x = p->x[i], y = p->y[i], z = p->z[i];
        const double Xp = c->p_X[0] * (x - c->position[0]) + c->p_X[1] * (y - c->position[1]) + c->p_X[2] * (z - c->position[2]);
        const double Yp = c->p_Y[0] * (x - c->position[0]) + c->p_Y[1] * (y - c->position[1]) + c->p_Y[2] * (z - c->position[2]);
        const cplx arg = t - c->t_peak;
        const cplx stc_exponent = 1. / stretch_factor * inv_tau2 * (arg * arg);
        const cplx stcfactor = prefactor * std::exp(-stc_exponent);
        const cplx exp_argument = -(Xp * Xp + Yp * Yp) * inv_complex_waist_2;
        const double amplitude = (stcfactor * std::exp(exp_argument)).real();
        const double sign_charge = (p->w[i] > 0) ? -1 : 1;
        const double v_over_c = sign_charge * c->mobility * amplitude;
        double vx = PhysConst::c * v_over_c * c->p_X[0];
        double vy = PhysConst::c * v_over_c * c->p_X[1];
        double vz = PhysConst::c * v_over_c * c->p_X[2];
        // When running in the boosted-frame, their is additional velocity along nvec (:907-912)
        if (gamma_boost > 1.) {
            vx -= PhysConst::c * beta_boost * c->nvec[0];
            vy -= PhysConst::c * beta_boost * c->nvec[1];
            vz -= PhysConst::c * beta_boost * c->nvec[2];
        }
        const double gamma = gamma_boost / std::sqrt(1. - v_over_c * v_over_c);
        p->ux[i] = gamma * vx; p->uy[i] = gamma * vy; p->uz[i] = gamma * vz;
        p->x[i] = x + vx * dt; p->y[i] = y + vy * dt; p->z[i] = z + vz * dt;
    }
    return 0;
}

int orc_apply_pec_e(const wxa_field_view E[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                    const int32_t pec_lo[3], const int32_t pec_hi[3], const int32_t ng[3], void*) {
    return apply_pec<true>(E, dom_lo, dom_hi, pec_lo, pec_hi, ng);
}
int orc_apply_pec_b(const wxa_field_view B[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                    const int32_t pec_lo[3], const int32_t pec_hi[3], const int32_t ng[3], void*) {
    return apply_pec<false>(B, dom_lo, dom_hi, pec_lo, pec_hi, ng);
}

// FillBoundaryAndSync's extra step (Communication.cpp:99-101,109-110): shared nodal
// points take their owner's value; on a single periodic brick the high-edge nodal
// point duplicates the low-edge one.
int orc_sync_nodal_periodic(const wxa_field_view* f, const int periodic[3], void*) {
    const Arr a(*f);
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d] || !f->stag[d]) continue;
        const int nc = ncell_of(*f, d);
        int rl[3], rh[3];
        for (int e = 0; e < 3; ++e) { rl[e] = vlo(*f, e); rh[e] = vhi(*f, e); }
        rl[d] = vlo(*f, d) + nc; rh[d] = rl[d] + 1;
        for (int k = rl[2]; k < rh[2]; ++k)
            for (int j = rl[1]; j < rh[1]; ++j)
                for (int i = rl[0]; i < rh[0]; ++i) {
                    int s[3] = {i, j, k};
                    s[d] -= nc;
                    a(i, j, k) = a(s[0], s[1], s[2]);
                }
    }
    return 0;
}

// amrex FabArray::SumBoundary(src_ng, dst_ng = all) on a single periodic brick:
// every point = sum over periodic images of the values in valid + src_ng guards
// (Source/Parallelization/WarpXSumGuardCells.cpp:17-37).
int orc_sum_boundary_periodic(const wxa_field_view* f, const int src_ng[3], const int periodic[3], void*) {
    const Arr a(*f);
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d]) continue;
        const int nc = ncell_of(*f, d);
        const int a0 = f->lo[d], a1 = f->lo[d] + f->n[d];
        const int s0 = vlo(*f, d) - src_ng[d], s1 = vhi(*f, d) + src_ng[d];
        // the lines along d are independent of each other: all cores (each line is summed in the same order as before);
        // threads split the slowest of the other two dimensions and keep whole rows (no shared cache lines)
        const int d2 = d == 2 ? 1 : 2, d1 = d == 0 ? 1 : 0;
        auto plane = [&](const int u, std::vector<double>& line) {
            line.resize(f->n[d]);
            for (int v = f->lo[d1]; v < f->lo[d1] + f->n[d1]; ++v) {
                int idx[3];
                idx[d1] = v; idx[d2] = u;
                for (int t = a0; t < a1; ++t) { idx[d] = t; line[t - a0] = a(idx[0], idx[1], idx[2]); }
                for (int t = a0; t < a1; ++t) {
                    double sum = 0.0;
                    // images t + m*nc inside the source range, ascending
                    const int first = s0 + (((t - s0) % nc) + nc) % nc;
                    for (int s = first; s < s1; s += nc) sum += line[s - a0];
                    idx[d] = t;
                    a(idx[0], idx[1], idx[2]) = sum;
                }
            }
        };
        if ((long)f->n[0] * f->n[1] * f->n[2] >= 262144) {   // small arrays stay serial, see orc_fill_boundary_periodic
#pragma omp parallel
            {
                std::vector<double> line;
#pragma omp for
                for (int u = f->lo[d2]; u < f->lo[d2] + f->n[d2]; ++u) plane(u, line);
            }
        } else {
            std::vector<double> line;
            for (int u = f->lo[d2]; u < f->lo[d2] + f->n[d2]; ++u) plane(u, line);
        }
    }
    return 0;
}

int orc_field_set_zero(const wxa_field_view* f, void*) {
    std::memset(f->p, 0, sizeof(double) * (size_t)f->kstride * f->n[2]);
    return 0;
}

// ---------------------------------------------------------------------------
// Host-side counterparts of the exchange helpers, used only when the test-suite runs the
// product's C++ host layer on CPU (tests/host_cpu) to exercise the multi-brick logic.
int orc_pack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], double* buf, void*) {
    const Arr a(*f);
    int64_t t = 0;
    for (int k = blo[2]; k < bhi[2]; ++k)
        for (int j = blo[1]; j < bhi[1]; ++j)
            for (int i = blo[0]; i < bhi[0]; ++i) buf[t++] = a(i, j, k);
    return 0;
}

int orc_unpack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], const double* buf,
                   int mode, void*) {
    const Arr a(*f);
    int64_t t = 0;
    for (int k = blo[2]; k < bhi[2]; ++k)
        for (int j = blo[1]; j < bhi[1]; ++j)
            for (int i = blo[0]; i < bhi[0]; ++i) {
                if (mode == 0) a(i, j, k) = buf[t++];
                else a(i, j, k) += buf[t++];
            }
    return 0;
}

// warpx.do_single_precision_comms: comm_float_type on the wire (ablastr/utils/Communication.cpp:37-56,90-106,159-170)
int orc_pack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], float* buf, void*) {
    const Arr a(*f);
    int64_t t = 0;
    for (int k = blo[2]; k < bhi[2]; ++k)
        for (int j = blo[1]; j < bhi[1]; ++j)
            for (int i = blo[0]; i < bhi[0]; ++i) buf[t++] = static_cast<float>(a(i, j, k));
    return 0;
}

int orc_unpack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], const float* buf,
                       int mode, void*) {
    const Arr a(*f);
    int64_t t = 0;
    for (int k = blo[2]; k < bhi[2]; ++k)
        for (int j = blo[1]; j < bhi[1]; ++j)
            for (int i = blo[0]; i < bhi[0]; ++i) {
                if (mode == 0) a(i, j, k) = static_cast<double>(buf[t++]);
                else a(i, j, k) += static_cast<double>(buf[t++]);
            }
    return 0;
}

int orc_partition_particles(const wxa_particle_view* src, const wxa_particle_view* dst, int dim, double lo,
                            double hi, int64_t counts[3], void*, void*) {
    const double* pos = dim == 0 ? src->x : (dim == 1 ? src->y : src->z);
    const double* s[7] = {src->x, src->y, src->z, src->w, src->ux, src->uy, src->uz};
    double* d[7] = {dst->x, dst->y, dst->z, dst->w, dst->ux, dst->uy, dst->uz};
    counts[0] = counts[1] = counts[2] = 0;
    for (int64_t i = 0; i < src->np; ++i) counts[pos[i] < lo ? 1 : (pos[i] >= hi ? 2 : 0)]++;
    int64_t cur[3] = {0, counts[0], counts[0] + counts[1]};
    for (int64_t i = 0; i < src->np; ++i) {
        const int key = pos[i] < lo ? 1 : (pos[i] >= hi ? 2 : 0);
        const int64_t t = cur[key]++;
        for (int c = 0; c < 7; ++c) d[c][t] = s[c][i];
        if (src->idcpu && dst->idcpu) dst->idcpu[t] = src->idcpu[i];
    }
    return 0;
}

// stable counting sort by cell (i fastest); any grouping by cell is a valid SortParticlesByBin.
// Retired particles (idcpu == WXA_IDCPU_RETIRED, see orc_pack_leavers) go behind the live ones;
// if ws is given it receives the live count (one int64).
int orc_sort_particles_by_cell(const wxa_particle_view* src, const wxa_particle_view* dst, const double plo[3],
                               const double dinv[3], const int32_t*, const int32_t ncell[3], void* ws, void*) {
    const int64_t nc = (int64_t)ncell[0] * ncell[1] * ncell[2];
    std::vector<int64_t> off(nc + 2, 0);
    std::vector<int64_t> key(src->np);
    for (int64_t p = 0; p < src->np; ++p) {
        int c[3];
        const double pos[3] = {src->x[p], src->y[p], src->z[p]};
        for (int d = 0; d < 3; ++d) {
            c[d] = (int)std::floor((pos[d] - plo[d]) * dinv[d]);
            c[d] = std::min(std::max(c[d], 0), ncell[d] - 1);
        }
        key[p] = c[0] + (int64_t)ncell[0] * (c[1] + (int64_t)ncell[1] * c[2]);
        if (src->idcpu && src->idcpu[p] == WXA_IDCPU_RETIRED) key[p] = nc;
        off[key[p] + 1]++;
    }
    for (int64_t c = 0; c <= nc; ++c) off[c + 1] += off[c];
    if (ws) *static_cast<int64_t*>(ws) = off[nc];
    const double* s[7] = {src->x, src->y, src->z, src->w, src->ux, src->uy, src->uz};
    double* d[7] = {dst->x, dst->y, dst->z, dst->w, dst->ux, dst->uy, dst->uz};
    for (int64_t p = 0; p < src->np; ++p) {
        const int64_t t = off[key[p]]++;
        for (int c = 0; c < 7; ++c) d[c][t] = s[c][p];
        if (src->idcpu && dst->idcpu) dst->idcpu[t] = src->idcpu[p];
    }
    return 0;
}

int orc_sort_live_count(void* ws, int64_t* n, void*) {
    if (!ws || !n) return -1;
    *n = *static_cast<int64_t*>(ws);
    return 0;
}

// WarpXParticleContainer::ApplyBoundaryConditions (Source/Particles/WarpXParticleContainer.cpp:1574-1660)
// with apply_boundary / apply_boundaries (Source/Particles/ParticleBoundaries_K.H:20-175): reflecting and
// absorbing (reflection probability 0) walls.  A lost particle is retired in place (the reference
// invalidates its id and lets Redistribute drop it).
int orc_apply_particle_boundaries(const wxa_particle_view* p, const double prob_lo[3], const double prob_hi[3],
                                  const int32_t bc_lo[3], const int32_t bc_hi[3], int64_t* n_lost, void*, void*) {
    double* pos[3] = {p->x, p->y, p->z};
    double* u[3] = {p->ux, p->uy, p->uz};
    int64_t lost_total = 0;
    for (int64_t ip = 0; ip < p->np; ++ip) {
        if (p->idcpu[ip] == WXA_IDCPU_RETIRED) {   // :1617-1618 skip particles already flagged; this library keeps
            // them in the tile until the next sort (still pushed, weight 0): parked inside the domain, momentum 0
            for (int d = 0; d < 3; ++d) {
                pos[d][ip] = std::min(std::max(pos[d][ip], prob_lo[d]), std::nextafter(prob_hi[d], prob_lo[d]));
                u[d][ip] = 0.0;
            }
            continue;
        }
        bool lost = false, flip[3] = {false, false, false};
        for (int d = 0; d < 3; ++d) {
            double& x = pos[d][ip];
            if (x < prob_lo[d]) {
                if (bc_lo[d] == WXA_PBOUNDARY_ABSORBING) lost = true;
                else if (bc_lo[d] == WXA_PBOUNDARY_REFLECTING) { x = 2 * prob_lo[d] - x; flip[d] = true; }
            } else if (x > prob_hi[d]) {
                if (bc_hi[d] == WXA_PBOUNDARY_ABSORBING) lost = true;
                else if (bc_hi[d] == WXA_PBOUNDARY_REFLECTING) { x = 2 * prob_hi[d] - x; flip[d] = true; }
            }
        }
        if (lost) {
            for (int d = 0; d < 3; ++d) {
                pos[d][ip] = std::min(std::max(pos[d][ip], prob_lo[d]), std::nextafter(prob_hi[d], prob_lo[d]));
                u[d][ip] = 0.0;
            }
            p->w[ip] = 0.0;
            p->idcpu[ip] = WXA_IDCPU_RETIRED;
            ++lost_total;
        } else {
            for (int d = 0; d < 3; ++d)
                if (flip[d]) u[d][ip] = -u[d][ip];   // ParticleBoundaries_K.H:160-170 (reflect_all_velocities off)
        }
    }
    if (n_lost) *n_lost = lost_total;
    return 0;
}

// The brick-to-brick part of amrex ParticleContainer::Redistribute as the host layer drives it
// (include/warpx_amd.h, "Redistribute without moving the tile"): periodic wrap plus the lists of
// particles that left the brick, by the first split direction in which they are outside (decided
// on the unwrapped position).
int orc_wrap_and_classify(const wxa_particle_view* p, int64_t first, int64_t count, const double prob_lo[3],
                          const double prob_hi[3], const int periodic[3], const double brick_lo[3],
                          const double brick_hi[3], const int split[3], int32_t* lists, int64_t capacity,
                          int64_t counts[6], void*, void*) {
    for (int c = 0; c < 6; ++c) counts[c] = 0;
    double* pos[3] = {p->x, p->y, p->z};
    for (int64_t ip = first; ip < first + count; ++ip) {
        int code = -1;
        for (int d = 0; d < 3 && code < 0; ++d)
            if (split[d]) code = pos[d][ip] < brick_lo[d] ? 2 * d : (pos[d][ip] >= brick_hi[d] ? 2 * d + 1 : -1);
        if (code >= 0 && p->idcpu[ip] == WXA_IDCPU_RETIRED) {
            // parked again on the brick's side of the faces (mirror of wrap_classify_kernel)
            code = -1;
            for (int d = 0; d < 3; ++d)
                if (split[d])
                    pos[d][ip] = std::min(std::max(pos[d][ip], brick_lo[d]), std::nextafter(brick_hi[d], brick_lo[d]));
        }
        if (code >= 0) {
            if (counts[code] < capacity) lists[code * capacity + counts[code]] = (int32_t)ip;
            counts[code]++;
        }
    }
    wxa_particle_view r = *p;
    r.x += first; r.y += first; r.z += first;
    r.np = count;
    return orc_enforce_periodic(&r, prob_lo, prob_hi, periodic, nullptr);
}

// The same scan with the leavers listed by destination brick: list (ox + 1) + 3 (oy + 1) + 9 (oz + 1)
// (include/warpx_amd.h, wxa_wrap_and_classify_dest)
int orc_wrap_and_classify_dest(const wxa_particle_view* p, int64_t first, int64_t count, const double prob_lo[3],
                               const double prob_hi[3], const int periodic[3], const double brick_lo[3],
                               const double brick_hi[3], const int split[3], int32_t* lists, int64_t capacity,
                               int64_t counts[27], void*, void*) {
    for (int c = 0; c < 27; ++c) counts[c] = 0;
    double* pos[3] = {p->x, p->y, p->z};
    for (int64_t ip = first; ip < first + count; ++ip) {
        int o[3] = {0, 0, 0};
        for (int d = 0; d < 3; ++d)
            if (split[d]) o[d] = pos[d][ip] < brick_lo[d] ? -1 : (pos[d][ip] >= brick_hi[d] ? 1 : 0);
        int code = (o[0] + 1) + 3 * (o[1] + 1) + 9 * (o[2] + 1);
        if (code != 13 && p->idcpu[ip] == WXA_IDCPU_RETIRED) {
            code = 13;   // parked again on the brick's side of the faces (mirror of wrap_classify_kernel)
            for (int d = 0; d < 3; ++d)
                if (split[d])
                    pos[d][ip] = std::min(std::max(pos[d][ip], brick_lo[d]), std::nextafter(brick_hi[d], brick_lo[d]));
        }
        if (code != 13) {
            if (counts[code] < capacity) lists[code * capacity + counts[code]] = (int32_t)ip;
            counts[code]++;
        }
    }
    wxa_particle_view r = *p;
    r.x += first; r.y += first; r.z += first;
    r.np = count;
    return orc_enforce_periodic(&r, prob_lo, prob_hi, periodic, nullptr);
}

int orc_pack_leavers(const wxa_particle_view* p, const int32_t* list, int64_t n, void* msg, int64_t row_len,
                     int64_t offset, int retire, const double brick_lo[3], const double brick_hi[3], void*) {
    double* m = static_cast<double*>(msg) + offset;
    const double* s[7] = {p->x, p->y, p->z, p->w, p->ux, p->uy, p->uz};
    double* pos[3] = {p->x, p->y, p->z};
    for (int64_t t = 0; t < n; ++t) {
        const int64_t ip = list[t];
        for (int c = 0; c < 7; ++c) m[c * row_len + t] = s[c][ip];
        reinterpret_cast<uint64_t*>(m)[7 * row_len + t] = p->idcpu[ip];
        if (retire) {
            for (int d = 0; d < 3; ++d)
                pos[d][ip] = std::min(std::max(pos[d][ip], brick_lo[d]), std::nextafter(brick_hi[d], brick_lo[d]));
            p->w[ip] = 0.0; p->ux[ip] = 0.0; p->uy[ip] = 0.0; p->uz[ip] = 0.0;
            p->idcpu[ip] = WXA_IDCPU_RETIRED;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Diagnostics that define the parity metric.

// MultiFab::norm2(0, periodicity)^2: sum of squares over unique points
// (Source/Diagnostics/ReducedDiags/FieldEnergy.cpp:121-129): nodal duplicates on
// the high periodic edge are counted once.
double orc_sum_sq_unique(const wxa_field_view* f) {
    const Arr a(*f);
    long double s = 0.0L;
    int hi[3];
    for (int d = 0; d < 3; ++d) hi[d] = vlo(*f, d) + ncell_of(*f, d);
    for (int k = vlo(*f, 2); k < hi[2]; ++k)
        for (int j = vlo(*f, 1); j < hi[1]; ++j)
            for (int i = vlo(*f, 0); i < hi[0]; ++i) s += (long double)a(i, j, k) * a(i, j, k);
    return (double)s;
}

// FieldEnergy.cpp:146-151: out[0] = total, out[1] = E part, out[2] = B part
void orc_field_energy(const wxa_field_view E[3], const wxa_field_view B[3], const double dx[3],
                      double out[3]) {
    const double dV = dx[0] * dx[1] * dx[2];
    const double Es = orc_sum_sq_unique(&E[0]) + orc_sum_sq_unique(&E[1]) + orc_sum_sq_unique(&E[2]);
    const double Bs = orc_sum_sq_unique(&B[0]) + orc_sum_sq_unique(&B[1]) + orc_sum_sq_unique(&B[2]);
    out[1] = 0.5 * Es * PhysConst::ep0 * dV;
    out[2] = 0.5 * Bs / PhysConst::mu0 * dV;
    out[0] = out[1] + out[2];
}

// Source/Diagnostics/ReducedDiags/ParticleEnergy.cpp:95-200 with
// Source/Particles/Algorithms/KineticEnergy.H:31-45: sum_p w * m u^2 / (1 + gamma)
double orc_particle_energy(const wxa_particle_view* p, double mass) {
    constexpr double inv_c2 = 1.0 / (PhysConst::c * PhysConst::c);
    long double s = 0.0L;
    for (int64_t i = 0; i < p->np; ++i) {
        const double u2 = p->ux[i] * p->ux[i] + p->uy[i] * p->uy[i] + p->uz[i] * p->uz[i];
        const double gamma = std::sqrt(1.0 + u2 * inv_c2);
        s += (long double)p->w[i] * (1.0 / (1.0 + gamma) * mass * u2);
    }
    return (double)s;
}

// Source/Diagnostics/ReducedDiags/ParticleMomentum.cpp: sum_p w * m * u
void orc_particle_momentum(const wxa_particle_view* p, double mass, double out[3]) {
    long double s[3] = {0, 0, 0};
    for (int64_t i = 0; i < p->np; ++i) {
        s[0] += (long double)p->w[i] * mass * p->ux[i];
        s[1] += (long double)p->w[i] * mass * p->uy[i];
        s[2] += (long double)p->w[i] * mass * p->uz[i];
    }
    out[0] = (double)s[0]; out[1] = (double)s[1]; out[2] = (double)s[2];
}

// The reductions of the reduced diagnostics with the product's signatures (include/warpx_amd.h: wxa_reduce_field,
// wxa_reduce_particles), so that the host layer's ReducedDiags can run on this backend and the HIP kernels can be
// checked against it.  Plain loops in long double: FieldEnergy.cpp:81-157 (norm2 / norminf over the box the caller
// passes), ParticleEnergy.cpp:95-200 + KineticEnergy.H:33-67, ParticleMomentum.cpp:122-253, ParticleNumber.cpp:97-139.
int orc_reduce_field(const wxa_field_view* f, const int32_t lo[3], const int32_t hi[3], double* sum_sq, double* max_abs,
                     void*) {
    if (!f || !f->p || !lo || !hi) return -1;
    const Arr a(*f);
    long double s = 0.0L;
    double m = 0.0;
    for (int k = lo[2]; k < hi[2]; ++k)
        for (int j = lo[1]; j < hi[1]; ++j)
            for (int i = lo[0]; i < hi[0]; ++i) {
                const double x = a(i, j, k);
                s += (long double)x * x;
                m = std::max(m, std::fabs(x));
            }
    if (sum_sq) *sum_sq = (double)s;
    if (max_abs) *max_abs = m;
    return 0;
}

int orc_reduce_particles(const wxa_particle_view* p, double mass, int32_t photon, double out[6], void*) {
    if (!p || !out) return -1;
    constexpr double inv_c2 = 1.0 / (PhysConst::c * PhysConst::c);
    constexpr double me_c = PhysConst::m_e * PhysConst::c;
    long double s[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < p->np; ++i) {
        if (p->idcpu && p->idcpu[i] == WXA_IDCPU_RETIRED) continue;
        const double w = p->w[i], ux = p->ux[i], uy = p->uy[i], uz = p->uz[i];
        const double u2 = ux * ux + uy * uy + uz * uz;
        double ekin;
        if (photon) {
            ekin = me_c * std::sqrt(u2);
        } else {
            const double gamma = std::sqrt(1.0 + u2 * inv_c2);
            ekin = 1.0 / (1.0 + gamma) * mass * u2;
        }
        s[0] += (long double)w * ekin;
        s[1] += w;
        s[2] += (long double)w * mass * ux;
        s[3] += (long double)w * mass * uy;
        s[4] += (long double)w * mass * uz;
        s[5] += 1;
    }
    for (int c = 0; c < 6; ++c) out[c] = (double)s[c];
    return 0;
}

// Plotfile/checksum view: each field interpolated to cell centres
// (Source/Diagnostics/ComputeDiagFunctors/CellCenterFunctor.cpp:20-29 ->
// Source/ablastr/coarsen/sample.H:30-95 with cr = 1, sc = cell), then sum |Q| over the
// covering grid (Regression/Checksum/checksum.py:62-217).
double orc_cell_centered_abs_sum(const wxa_field_view* f) {
    const Arr a(*f);
    int np[3];
    for (int l = 0; l < 3; ++l) np[l] = 1 + std::abs(f->stag[l] - 0);
    const double wx = 1.0 / np[0], wy = 1.0 / np[1], wz = 1.0 / np[2];
    long double s = 0.0L;
    const int i0 = vlo(*f, 0), j0 = vlo(*f, 1), k0 = vlo(*f, 2);
    for (int k = k0; k < k0 + ncell_of(*f, 2); ++k)
        for (int j = j0; j < j0 + ncell_of(*f, 1); ++j)
            for (int i = i0; i < i0 + ncell_of(*f, 0); ++i) {
                double c = 0.0;  // idx_min = ic - sc*(1-sf) = ic for sc = 0
                for (int kr = 0; kr < np[2]; ++kr)
                    for (int jr = 0; jr < np[1]; ++jr)
                        for (int ir = 0; ir < np[0]; ++ir) c += wx * wy * wz * a(i + ir, j + jr, k + kr);
                s += std::fabs(c);
            }
    return (double)s;
}

double orc_abs_sum(const double* v, int64_t n, double scale) {
    long double s = 0.0L;
    for (int64_t i = 0; i < n; ++i) s += std::fabs(v[i] * scale);
    return (double)s;
}

}  // extern "C"

// ===========================================================================
// Step-level oracle: single brick, fully periodic.
// ===========================================================================
namespace {

struct Field {
    std::vector<double> data;
    wxa_field_view v{};
    void alloc(const int ncell[3], const int stag[3], const int ng[3]) {
        for (int d = 0; d < 3; ++d) {
            v.lo[d] = -ng[d]; v.ng[d] = ng[d]; v.stag[d] = stag[d];
            v.n[d] = ncell[d] + stag[d] + 2 * ng[d];
        }
        v.jstride = v.n[0];
        v.kstride = v.jstride * v.n[1];
        data.assign((size_t)v.kstride * v.n[2], 0.0);
        v.p = data.data();
    }
};

struct Species {
    double q, m;
    std::vector<double> a[7];
    std::vector<uint64_t> id;
    // <species>.do_continuous_injection with a NUniformPerCell constant-density injector
    bool inject = false;
    wxa_plasma_injector inj{};
    double inj_pos = 0.0;   // WarpXParticleContainer::m_current_injection_position
    double ext_eb[6] = {0, 0, 0, 0, 0, 0};   // m_E_external_particle, m_B_external_particle
    bool do_crr = false;                     // do_classical_radiation_reaction
    wxa_particle_view view() {
        wxa_particle_view p{};
        p.x = a[0].data(); p.y = a[1].data(); p.z = a[2].data(); p.w = a[3].data();
        p.ux = a[4].data(); p.uy = a[5].data(); p.uz = a[6].data();
        p.idcpu = id.empty() ? nullptr : id.data();
        p.np = (int64_t)a[0].size();
        return p;
    }
};

}  // namespace

// LaserParticleContainer (Source/Particles/LaserParticleContainer.cpp): pairs of +-weight macro-particles
// on the antenna plane, moved with a prescribed velocity proportional to the field to emit, depositing
// current like any other species (charge 1, infinite mass).
struct LaserAntenna {
    wxa_laser_antenna cfg{};
    double nvec[3], p_X[3], p_Y[3];   // plane normal, polarization, second polarization vector
    double position[3];
    double Z0_lab = 0.0;              // boosted frame: the antenna plane's lab-frame position along the boost (:190)
    double S_X = 0, S_Y = 0, mobility = 0, weight = 0;
    Species parts;                    // q = 1 (:86), m irrelevant (never pushed by the fields)
};

struct orc_sim {
    wxa_sim_config cfg;
    double dx[3], dinv[3], dt;
    double ckc_x[5] = {0}, ckc_y[5] = {0}, ckc_z[5] = {0};   // algo.maxwell_solver = ckc
    int ng_EB[3], ng_J[3], ng_depos_J[3], ng_gather[3], ng_solver[3], ng_rho[3];
    Field E[3], B[3], J[3], Jtmp, rho;
    // particles.use_fdtd_nci_corr (WarpX::InitNCICorrector, WarpXInitData.cpp:858-890): filtered copies of E and B
    Field nciE[3], nciB[3];
    double nci_exeybz[5] = {0.5, 0, 0, 0, 0}, nci_bxbyez[5] = {0.5, 0, 0, 0, 0};
    std::vector<std::unique_ptr<Species>> species;
    bool is_synchronized = true;
    int64_t istep = 0;
    double cur_time = 0.0;
    double timers[8] = {0};
    int64_t counts[8] = {0};
    bool do_timers = false;

    wxa_field_view Ev[3], Bv[3], Jv[3];
    int periodic[3] = {1, 1, 1};
    // boundary.field_lo/hi = pec (fields only)
    bool any_pec = false;
    int32_t pec_lo[3] = {0, 0, 0}, pec_hi[3] = {0, 0, 0}, dom_lo[3] = {0, 0, 0}, dom_hi[3] = {0, 0, 0};
    int32_t ng_gather32[3] = {0, 0, 0};
    // boundary.particle_lo/hi resolved (WXA_PBOUNDARY_ABSORBING / _REFLECTING / _PERIODIC)
    int32_t pbc_lo[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    int32_t pbc_hi[3] = {WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC, WXA_PBOUNDARY_PERIODIC};
    bool any_particle_wall = false;
    // current physical domain (moves with the moving window)
    double plo[3] = {0, 0, 0}, phi[3] = {0, 0, 0};
    // warpx.do_moving_window
    // warpx.gamma_boost, boost along z (WarpXUtil.cpp:114-141)
    double gamma_boost = 1.0, beta_boost = 0.0;
    bool mw_on = false;
    int mw_dir = 2;
    double mw_v = 0.0;   // m/s
    double mw_x = 0.0;   // WarpX::moving_window_x
    std::vector<std::unique_ptr<struct LaserAntenna>> lasers;
    // <diag>.diag_type = BackTransformed, fields (BTDiagnostics.cpp; see btd_compute_and_pack below)
    struct BtdSnapshot {
        double t_lab = 0, zlo_lab = 0, zhi_lab = 0, z_boost = 0, z_lab = 0;
        int ksmall = 0, kbig = 0, counter = 0, last_valid = 0, full = 0;
        int n[3] = {0, 0, 0};
        std::vector<double> data;   // [comp][k][j][i]
        std::vector<std::array<std::vector<double>, 7>> particles;   // per species: x y z w ux uy uz in the lab frame
    };
    std::vector<BtdSnapshot> btd;
    bool btd_write_species = false;
    int btd_buffer_size = 0;
    double btd_dt_snap = 0.0;
    // warpx.reduced_diags_names: FieldEnergy / ParticleEnergy / ParticleMomentum / ParticleNumber (rd_compute below)
    struct ReducedDiag {
        std::string name, type, file;
        int start = 0, stop = 2147483647, period = 1;   // one slice "start:stop:period" (or "period")
        std::vector<double> data;
    };
    std::vector<ReducedDiag> rdiags;
    bool rdiags_started = false;

    wxa_grid_geom geom_for(const int ng[3]) const {
        // WarpX::LowerCorner(box.grow(ng)) = prob_lo + box.lo * dx (Source/WarpX.cpp:2851-2875)
        wxa_grid_geom g{};
        for (int d = 0; d < 3; ++d) {
            g.lo[d] = -ng[d];
            g.xyzmin[d] = plo[d] + (double)(-ng[d]) * dx[d];
            g.dinv[d] = dinv[d];
        }
        return g;
    }
};

namespace {

double now_ms() {
#ifdef _OPENMP
    return omp_get_wtime() * 1e3;
#else
    return 0.0;
#endif
}
struct Tic {
    orc_sim* s; int id; double t0;
    Tic(orc_sim* s_, int id_) : s(s_), id(id_), t0(s_->do_timers ? now_ms() : 0.0) {}
    ~Tic() { if (s->do_timers) { s->timers[id] += now_ms() - t0; s->counts[id]++; } }
};


// ---- plasma injection: PhysicalParticleContainer::AddPlasma (PhysicalParticleContainer.cpp:924-1333)
// restricted to NUniformPerCell, constant density, at rest, lab frame, one box ------------------------
void add_plasma(orc_sim* s, Species& sp, const double part_lo[3], const double part_hi[3]) {
    const wxa_plasma_injector& in = sp.inj;
    // find_overlap (Source/Particles/AddPlasmaUtilities.cpp:12-43) with tile_realbox = the whole domain
    double olo[3], ohi[3];
    int nov[3];
    for (int d = 0; d < 3; ++d) {
        const double tlo = s->plo[d], thi = s->phi[d], dx = s->dx[d];
        if (!(tlo <= part_hi[d])) return;
        olo[d] = part_lo[d] + std::max(std::floor((tlo - part_lo[d]) / dx), 0.0) * dx;
        if (!(thi >= part_lo[d])) return;
        ohi[d] = part_hi[d] - std::max(std::floor((part_hi[d] - thi) / dx), 0.0) * dx;
        nov[d] = (int)std::round((ohi[d] - olo[d]) / dx);   // cells 0 .. nov-1
    }
    const int nppc = in.ppc[0] * in.ppc[1] * in.ppc[2];
    if (s->gamma_boost > 1.0) {
        // the boosted branch of AddPlasma (:1021-1022 ballistic correction of the cell test, :1181-1247 lab-frame
        // bounds, transformed density and momentum) lives in the exported routine below; plasma at rest in the lab
        wxa_plasma_injector inj = in;
        inj.gamma_boost = s->gamma_boost;
        inj.t = s->cur_time;                       // warpx.gett_new(lev)
        const int64_t room = (int64_t)nov[0] * nov[1] * nov[2] * nppc;
        if (room <= 0) return;
        std::vector<double> col[7];
        for (auto& c : col) c.assign((size_t)room, 0.0);
        wxa_particle_view dst{};
        dst.x = col[0].data(); dst.y = col[1].data(); dst.z = col[2].data(); dst.w = col[3].data();
        dst.ux = col[4].data(); dst.uy = col[5].data(); dst.uz = col[6].data(); dst.idcpu = nullptr; dst.np = room;
        const int32_t nc[3] = {nov[0], nov[1], nov[2]};
        int64_t added = 0;
        orc_add_plasma(&dst, &inj, olo, nc, s->dx, s->plo, s->phi, nullptr, &added, nullptr, nullptr);
        for (int c = 0; c < 7; ++c) sp.a[c].insert(sp.a[c].end(), col[c].begin(), col[c].begin() + added);
        if (!sp.id.empty() || s->any_particle_wall) sp.id.resize(sp.a[0].size(), 0);
        return;
    }
    const double scale_fac = s->dx[0] * s->dx[1] * s->dx[2] / nppc;   // compute_scale_fac_volume
    auto inside = [&](double x, double y, double z) {   // InjectorPosition::insideBounds
        return x < in.hi[0] && x >= in.lo[0] && y < in.hi[1] && y >= in.lo[1] && z < in.hi[2] && z >= in.lo[2];
    };
    for (int k = 0; k < nov[2]; ++k)
        for (int j = 0; j < nov[1]; ++j)
            for (int i = 0; i < nov[0]; ++i) {
                const int iv[3] = {i, j, k};
                double lo[3], hi[3];
                for (int d = 0; d < 3; ++d) { lo[d] = olo[d] + (iv[d] + 0.0) * s->dx[d]; hi[d] = olo[d] + (iv[d] + 1.0) * s->dx[d]; }
                // InjectorPosition::overlapsWith (:225-233): the cell overlaps the plasma region
                bool overlaps = true;
                for (int d = 0; d < 3; ++d) overlaps = overlaps && !(lo[d] > in.hi[d] || hi[d] < in.lo[d]);
                if (!overlaps) continue;
                // :1030-1048 corners or centre with non-zero density (constant density: inside bounds)
                bool any = false;
                for (int a = 0; a < 3 && !any; ++a)
                    for (int b = 0; b < 3 && !any; ++b)
                        for (int c = 0; c < 3 && !any; ++c) {
                            const double x = a == 0 ? lo[0] : (a == 1 ? (lo[0] + hi[0]) / 2. : hi[0]);
                            const double y = b == 0 ? lo[1] : (b == 1 ? (lo[1] + hi[1]) / 2. : hi[1]);
                            const double z = c == 0 ? lo[2] : (c == 1 ? (lo[2] + hi[2]) / 2. : hi[2]);
                            any = inside(x, y, z) && in.density > 0;
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
                    for (int d = 0; d < 3; ++d) pos[d] = olo[d] + (iv[d] + r[d]) * s->dx[d];   // getCellCoords
                    // tile_realbox.contains (amrex RealBox::contains: strictly inside)
                    bool in_tile = true;
                    for (int d = 0; d < 3; ++d) in_tile = in_tile && pos[d] > s->plo[d] && pos[d] < s->phi[d];
                    if (!in_tile || !inside(pos[0], pos[1], pos[2])) continue;
                    for (int d = 0; d < 3; ++d) sp.a[d].push_back(pos[d]);
                    sp.a[3].push_back(in.density * scale_fac);
                    for (int d = 4; d < 7; ++d) sp.a[d].push_back(0.0);
                    if (!sp.id.empty() || s->any_particle_wall) sp.id.resize(sp.a[0].size(), 0);
                }
            }
}

// ---- moving window: WarpX::shiftMF through the exported kernel (orc_shift_field_window below) ----
void shift_field(orc_sim* s, Field& f, int num_shift, int dir) {
    std::vector<double> tmp(f.data.size());
    orc_shift_field_window(&f.v, tmp.data(), dir, num_shift, s->periodic, nullptr);
}

// WarpX::MoveWindow (:138-476): forward window, lab frame, plasma at rest
int move_window(orc_sim* s, bool move_j) {
    if (!s->mw_on) return 0;
    const double c = PhysConst::c;
    s->mw_x += (s->mw_v - s->beta_boost * c) / (1 - s->mw_v * s->beta_boost / c) * s->dt;   // :157
    const int dir = s->mw_dir;
    // UpdateInjectionPosition (:60-136): the plasma (at rest in the lab: u_bulk = 0) drifts in a boosted frame and the
    // injection front with it, v' = (v - c beta) / (1 - v beta / c) along the boost direction (z)
    if (s->gamma_boost > 1.0 && dir == 2)
        for (auto& sp : s->species)
            if (sp->inject) sp->inj_pos += ((0.0 - c * s->beta_boost) / (1.0 - 0.0 * s->beta_boost / c)) * s->dt;
    const double cdx = s->dx[dir];
    const int num_shift = (int)((s->mw_x - s->plo[dir]) / cdx);   // :171
    if (num_shift == 0) return 0;
    s->plo[dir] += num_shift * cdx;                               // :181-184
    s->phi[dir] += num_shift * cdx;
    for (int c = 0; c < 3; ++c) {                                 // :222-246
        shift_field(s, s->B[c], num_shift, dir);
        shift_field(s, s->E[c], num_shift, dir);
        if (move_j) shift_field(s, s->J[c], num_shift, dir);
    }
    if (move_j) shift_field(s, s->rho, num_shift, dir);           // :365-375
    for (auto& sp : s->species) {                                 // :392-437 continuous injection
        if (!sp->inject) continue;
        const double new_pos = sp->inj_pos + std::floor((s->phi[dir] - sp->inj_pos) / cdx) * cdx;
        double blo[3] = {s->plo[0], s->plo[1], s->plo[2]}, bhi[3] = {s->phi[0], s->phi[1], s->phi[2]};
        blo[dir] = sp->inj_pos;
        bhi[dir] = new_pos;
        if (bhi[dir] > blo[dir] && sp->inj_pos != new_pos) {      // particleBox.ok()
            add_plasma(s, *sp, blo, bhi);
            sp->inj_pos = new_pos;
        }
    }
    return num_shift;
}

// ---- laser antenna --------------------------------------------------------------------------------
// LaserParticleContainer::InitData (:360-559), 3-D, one process
void laser_init(orc_sim* s, LaserAntenna& L) {
    // ComputeSpacing (:727-762): eps = dx * 1e-50 vanishes next to |u| = 1 or keeps dx/eps huge for |u| = 0
    const double eps = s->dx[0] * 1e-50;
    auto spacing = [&](const double u[3]) {
        return std::min(std::min(s->dx[0] / (std::abs(u[0]) + eps), s->dx[1] / (std::abs(u[1]) + eps)),
                        s->dx[2] / (std::abs(u[2]) + eps));
    };
    L.S_X = spacing(L.p_X);
    L.S_Y = spacing(L.p_Y);
    // ComputeWeightMobility (:764-781)
    L.mobility = 0.05 / L.cfg.e_max;
    L.weight = PhysConst::ep0 / L.mobility;
    L.weight *= L.S_X * L.S_Y;
    L.mobility = L.mobility / s->gamma_boost;   // :772-775 e_max is a lab-frame amplitude
    // plane index range from the corners of the injection box (:418-457), truncation towards zero
    int plo[2] = {INT_MAX, INT_MAX}, phi[2] = {INT_MIN, INT_MIN};
    for (int c = 0; c < 8; ++c) {
        const double pos[3] = {(c & 1) ? s->phi[0] : s->plo[0], (c & 2) ? s->phi[1] : s->plo[1],
                               (c & 4) ? s->phi[2] : s->plo[2]};
        double X = 0, Y = 0;
        for (int d = 0; d < 3; ++d) { X += L.p_X[d] * (pos[d] - L.position[d]); Y += L.p_Y[d] * (pos[d] - L.position[d]); }
        const int i = (int)(X / L.S_X), j = (int)(Y / L.S_Y);
        plo[0] = std::min(plo[0], i); plo[1] = std::min(plo[1], j);
        phi[0] = std::max(phi[0], i); phi[1] = std::max(phi[1], j);
    }
    // Box::next order: first index fastest (:503)
    for (int j = plo[1]; j <= phi[1]; ++j)
        for (int i = plo[0]; i <= phi[0]; ++i) {
            double pos[3];
            for (int d = 0; d < 3; ++d)   // Transform (:389-398)
                pos[d] = L.position[d] + (L.S_X * ((double)i + 0.5)) * L.p_X[d] + (L.S_Y * ((double)j + 0.5)) * L.p_Y[d];
            bool inside = true;           // RealBox::contains: strictly inside
            for (int d = 0; d < 3; ++d) inside = inside && pos[d] > s->plo[d] && pos[d] < s->phi[d];
            if (!inside) continue;
            for (int k = 0; k < 2; ++k) {
                for (int d = 0; d < 3; ++d) L.parts.a[d].push_back(pos[d]);
                L.parts.a[3].push_back(k == 0 ? L.weight : -L.weight);
                for (int d = 4; d < 7; ++d) L.parts.a[d].push_back(0.0);
            }
        }
    if (s->any_particle_wall) L.parts.id.assign(L.parts.a[0].size(), 0);
}

// LaserParticleContainer::Evolve (:563-713) through the exported kernel (orc_laser_push below)
void laser_push(orc_sim* s, LaserAntenna& L, double t, double dt) {
    wxa_laser_push_params par{};
    for (int d = 0; d < 3; ++d) { par.position[d] = L.position[d]; par.p_X[d] = L.p_X[d]; par.p_Y[d] = L.p_Y[d]; }
    par.mobility = L.mobility;
    par.e_max = L.cfg.e_max; par.wavelength = L.cfg.wavelength; par.waist = L.cfg.waist;
    par.duration = L.cfg.duration; par.t_peak = L.cfg.t_peak; par.focal_distance = L.cfg.focal_distance;
    for (int d = 0; d < 3; ++d) par.nvec[d] = L.nvec[d];
    par.gamma_boost = s->gamma_boost;
    // :574-579 the field to emit is the lab-frame one at the antenna's lab-frame time
    if (s->gamma_boost > 1.0) t = 1.0 / s->gamma_boost * t + s->beta_boost * L.Z0_lab / PhysConst::c;
    wxa_particle_view p = L.parts.view();
    orc_laser_push(&p, &par, t, dt, nullptr);
}

void fill_boundary_EB(orc_sim* s, wxa_field_view* F, const int ng[3], bool sync) {
    Tic t(s, 5);
    for (int c = 0; c < 3; ++c) {
        if (sync) orc_sync_nodal_periodic(&F[c], s->periodic, nullptr);
        orc_fill_boundary_periodic(&F[c], ng, s->periodic, nullptr);
    }
}

// Source/Particles/MultiParticleContainer.cpp:492-500 -> PhysicalParticleContainer::PushP
void push_p_all(orc_sim* s, double dt) {
    Tic t(s, 0);
    const wxa_grid_geom g = s->geom_for(s->ng_EB);
    for (auto& sp : s->species) {
        wxa_particle_view p = sp->view();
        orc_gather_push_ext(&p, s->Ev, s->Bv, &g, sp->q, sp->m, dt, s->cfg.nox, s->cfg.galerkin,
                            sp->do_crr ? WXA_PUSHER_BORIS_RR : s->cfg.particle_pusher, /*move=*/0, sp->ext_eb);
    }
}

// WarpX::OneStep_nosub (Source/Evolve/WarpXEvolve.cpp:354-455), FDTD branch
// BackTransformParticleFunctor::operator() (BackTransformParticleFunctor.cpp:76-152) right after the push of species
// `species` (the oracle keeps its arrays in order, so this could run at the end of the step like the reference's; it runs
// here to see the same domain as the product's host layer, see host/BTDiagnostics.hpp)
extern "C" int orc_btd_select_particles(const wxa_particle_view* p, const double* const old6[6], double z_boost,
                                        double z_boost_old, double t_boost, double dt, double t_lab, double gamma_boost,
                                        double* out, int64_t capacity, int64_t* n_selected, void*);
static void btd_pack_particles(orc_sim* s, int species, const wxa_particle_view& p, const std::vector<double> (&old_attr)[6],
                               double t_new, double dt) {
    const double g = s->gamma_boost, b = s->beta_boost, c = PhysConst::c;
    const double* old6[6];
    for (int k = 0; k < 6; ++k) old6[k] = old_attr[k].data();
    for (auto& sn : s->btd) {
        if ((int)sn.particles.size() <= species) sn.particles.resize((size_t)species + 1);
        const double zb = (sn.t_lab / g - t_new) * c / b, zl = (sn.t_lab - t_new / g) * c / b;
        const double cs = s->dx[2];
        const bool in_domain = !((zb <= s->plo[2] + 0.5 * cs) || (zb >= s->phi[2] - 0.5 * cs) || (zl <= sn.zlo_lab) ||
                                 (zl >= sn.zhi_lab));
        if (!in_domain || sn.full) continue;
        const double zb_old = (sn.t_lab / g - (t_new - dt)) * c / b;
        std::vector<double> out((size_t)7 * (size_t)p.np);
        int64_t n = 0;
        orc_btd_select_particles(&p, old6, zb, zb_old, t_new, dt, sn.t_lab, g, out.data(), p.np, &n, nullptr);
        for (int k = 0; k < 7; ++k)
            sn.particles[(size_t)species][(size_t)k].insert(sn.particles[(size_t)species][(size_t)k].end(),
                                                            out.begin() + (std::ptrdiff_t)((size_t)k * (size_t)p.np),
                                                            out.begin() + (std::ptrdiff_t)((size_t)k * (size_t)p.np + (size_t)n));
    }
}

void one_step_nosub(orc_sim* s) {
    const double dt = s->dt;
    // PushParticlesandDeposit (:1101-1180) -> MultiParticleContainer::Evolve (:460-482)
    for (int c = 0; c < 3; ++c) orc_field_set_zero(&s->Jv[c], nullptr);
    const wxa_grid_geom gEB = s->geom_for(s->ng_EB);
    const wxa_grid_geom gJ = s->geom_for(s->ng_depos_J);
    int species_index = -1;
    for (auto& sp : s->species) {
        ++species_index;
        wxa_particle_view p = sp->view();
        // CopyParticleAttribs (:2626-2629): the attributes before the push, for the back-transformed diagnostics
        std::vector<double> old_attr[6];
        const bool btd_particles = s->btd_write_species && !s->btd.empty() && p.np > 0;
        if (btd_particles) {
            const double* src[6] = {p.x, p.y, p.z, p.ux, p.uy, p.uz};
            for (int c = 0; c < 6; ++c) old_attr[c].assign(src[c], src[c] + p.np);
        }
        {   // PhysicalParticleContainer::Evolve :1961 PushPX
            Tic t(s, 0);
            const wxa_field_view* Eg = s->Ev;
            const wxa_field_view* Bg = s->Bv;
            wxa_field_view Ef[3], Bf[3];
            if (s->cfg.use_fdtd_nci_corr) {
                // :1900-1911 applyNCIFilter (:2097-2172): Ex, Ey, Bz with one stencil, Bx, By, Ez with the other, along z
                const double half[1] = {0.5};
                for (int c = 0; c < 3; ++c) { Ef[c] = s->nciE[c].v; Bf[c] = s->nciB[c].v; }
                orc_filter_stencil(&s->Ev[0], &Ef[0], half, 1, half, 1, s->nci_exeybz, 5, nullptr);
                orc_filter_stencil(&s->Ev[2], &Ef[2], half, 1, half, 1, s->nci_bxbyez, 5, nullptr);
                orc_filter_stencil(&s->Bv[1], &Bf[1], half, 1, half, 1, s->nci_bxbyez, 5, nullptr);
                orc_filter_stencil(&s->Ev[1], &Ef[1], half, 1, half, 1, s->nci_exeybz, 5, nullptr);
                orc_filter_stencil(&s->Bv[0], &Bf[0], half, 1, half, 1, s->nci_bxbyez, 5, nullptr);
                orc_filter_stencil(&s->Bv[2], &Bf[2], half, 1, half, 1, s->nci_exeybz, 5, nullptr);
                Eg = Ef; Bg = Bf;
            }
            orc_gather_push_ext(&p, Eg, Bg, &gEB, sp->q, sp->m, dt, s->cfg.nox, s->cfg.galerkin,
                                sp->do_crr ? WXA_PUSHER_BORIS_RR : s->cfg.particle_pusher, /*move=*/1, sp->ext_eb);
        }
        if (btd_particles) btd_pack_particles(s, species_index, p, old_attr, s->cur_time + dt, dt);
        {   // :2029-2038 DepositCurrent with relative_time = -0.5*dt
            Tic t(s, 1);
            orc_deposit_current(&p, s->Jv, &gJ, sp->q, dt, -0.5 * dt, s->cfg.nox, s->cfg.current_deposition,
                                nullptr, nullptr);
        }
    }
    for (auto& L : s->lasers) {   // LaserParticleContainer::Evolve (:563-713): lasers come after the species
        laser_push(s, *L, s->cur_time, dt);
        wxa_particle_view p = L->parts.view();
        orc_deposit_current(&p, s->Jv, &gJ, 1.0, dt, -0.5 * dt, s->cfg.nox, s->cfg.current_deposition, nullptr, nullptr);
    }
    {   // SyncCurrentAndRho (:583-652) -> SyncCurrent (WarpXComm.cpp:1073-1240)
        Tic t(s, 2);
        for (int c = 0; c < 3; ++c) {
            int src_ng[3];
            if (s->cfg.use_filter) {
                // ApplyFilterJ (WarpXComm.cpp:1357-1374): filter into a temp, copy back
                s->Jtmp.v = s->Jv[c];
                s->Jtmp.data.assign(s->J[c].data.size(), 0.0);
                s->Jtmp.v.p = s->Jtmp.data.data();
                orc_filter_bilinear(&s->Jv[c], &s->Jtmp.v, nullptr);
                std::memcpy(s->Jv[c].p, s->Jtmp.v.p, sizeof(double) * s->J[c].data.size());
            }
            // SumBoundaryJ (WarpXComm.cpp:1386-1424): ng_depos_J (+ stencil_length-1 if filtered), capped
            for (int d = 0; d < 3; ++d)
                src_ng[d] = std::min(s->ng_depos_J[d] + (s->cfg.use_filter ? 1 : 0), s->ng_J[d]);
            orc_sum_boundary_periodic(&s->Jv[c], src_ng, s->periodic, nullptr);
        }
        // :625-640 reflect the current density over PEC boundaries (ApplyJfieldBoundary)
        if (s->any_pec) orc_apply_pec_j(s->Jv, s->dom_lo, s->dom_hi, s->pec_lo, s->pec_hi, nullptr);
    }
    // WarpX::EvolveB / EvolveE end with ApplyBfieldBoundary / ApplyEfieldBoundary
    // (Source/FieldSolver/WarpXPushFieldsEM.cpp:926,990 -> WarpXFieldBoundaries.cpp:51-135)
    auto evolve_b = [&](double a_dt) {
        Tic t(s, 3);
        if (s->cfg.maxwell_solver == WXA_SOLVER_CKC) orc_evolve_b_ckc(s->Ev, s->Bv, a_dt, s->ckc_x, s->ckc_y, s->ckc_z, nullptr);
        else orc_evolve_b(s->Ev, s->Bv, a_dt, s->dinv, nullptr);
        if (s->any_pec) orc_apply_pec_b(s->Bv, s->dom_lo, s->dom_hi, s->pec_lo, s->pec_hi, s->ng_gather32, nullptr);
    };
    evolve_b(0.5 * dt);                                                         // :421
    fill_boundary_EB(s, s->Bv, s->ng_solver, true);                             // :422
    {   // :426
        Tic t(s, 4);
        orc_evolve_e(s->Ev, s->Bv, s->Jv, dt, s->dinv, nullptr);
        if (s->any_pec) orc_apply_pec_e(s->Ev, s->dom_lo, s->dom_hi, s->pec_lo, s->pec_hi, s->ng_gather32, nullptr);
    }
    fill_boundary_EB(s, s->Ev, s->ng_solver, true);                             // :433
    evolve_b(0.5 * dt);                                                         // :437
    // (:441-449 FillBoundaryB(ng_alloc_EB) only when safe_guard_cells or PML; skipped)
}

}  // namespace

extern "C" {

int orc_sim_create(const wxa_sim_config* cfg, const void* /*comm*/, orc_sim** out) {
    if (!cfg || !out) return -1;
    if (cfg->nox < 1 || cfg->nox > 4) return -1;
    if (cfg->nbricks[0] * cfg->nbricks[1] * cfg->nbricks[2] != 1) return -3;
    auto* s = new orc_sim();
    s->cfg = *cfg;
    if (cfg->gamma_boost > 1.0) {   // ReadBoostedFrameParameters (WarpXUtil.cpp:114-141)
        s->gamma_boost = cfg->gamma_boost;
        s->beta_boost = std::sqrt(1.0 - 1.0 / std::pow(cfg->gamma_boost, 2.0));
    }
    for (int d = 0; d < 3; ++d) {
        s->dx[d] = (cfg->prob_hi[d] - cfg->prob_lo[d]) / cfg->n_cell[d];
        s->plo[d] = cfg->prob_lo[d];
        s->phi[d] = cfg->prob_hi[d];
        s->dinv[d] = 1.0 / s->dx[d];
    }
    // Source/Evolve/WarpXComputeDt.cpp:41-102 + CartesianYeeAlgorithm::ComputeMaxDt (:48-56)
    const double* dx = s->dx;
    double deltat = cfg->cfl * 1.0 /
        (std::sqrt(1.0 / (dx[0] * dx[0]) + 1.0 / (dx[1] * dx[1]) + 1.0 / (dx[2] * dx[2])) * PhysConst::c);
    if (cfg->maxwell_solver == WXA_SOLVER_CKC) {   // CartesianCKCAlgorithm::ComputeMaxDt, InitializeStencilCoefficients
        if (cfg->grid_type != WXA_GRID_STAGGERED) { delete s; return -2; }
        deltat = cfg->cfl * orc_ckc_max_dt(dx);
        orc_ckc_stencil_coefficients(dx, s->ckc_x, s->ckc_y, s->ckc_z);
    } else if (cfg->maxwell_solver != WXA_SOLVER_YEE) { delete s; return -2; }
    s->dt = deltat;
    // Source/Parallelization/GuardCellManager.cpp:62-172,276-278,314-316
    const int nox = cfg->nox;
    for (int d = 0; d < 3; ++d) {
        const int ngt = nox;
        s->ng_EB[d] = (ngt % 2) ? ngt + 1 : ngt;
        int ngJ = ngt;
        ngJ += (int)std::ceil(PhysConst::c * 0.5 * s->dt / dx[d]);
        s->ng_depos_J[d] = ngJ;
        s->ng_J[d] = ngJ + (cfg->use_filter ? 1 : 0);
        s->ng_rho[d] = ngt + 1 + (int)std::ceil(PhysConst::c * s->dt / dx[d]);
        s->ng_gather[d] = std::min((nox + 1) / 2, s->ng_EB[d]);   // ng_FieldGather_noNCI (:314-316)
        if (d == 2 && cfg->use_fdtd_nci_corr) {
            // GuardCellManager.cpp:87-89: E and B carry the NCI stencil's cells in z; :319-325 the gather-depth exchange too
            const int ng = ngt + 4;   // NCIGodfreyFilter::m_stencil_width
            s->ng_EB[d] = (ng % 2) ? ng + 1 : ng;
            s->ng_gather[d] = std::min(s->ng_gather[d] + 4, s->ng_EB[d]);
        }
        s->ng_solver[d] = 1;
        // GuardCellManager.cpp:338: ng_FieldGather = max(ng_FieldGather, ng_FieldSolver)
        s->ng_gather[d] = std::max(s->ng_gather[d], s->ng_solver[d]);
        s->ng_gather32[d] = s->ng_gather[d];
        s->dom_lo[d] = 0;
        s->dom_hi[d] = cfg->n_cell[d] - 1;
        const int blo = cfg->field_boundary_lo[d], bhi = cfg->field_boundary_hi[d];
        if ((blo != WXA_BOUNDARY_PERIODIC && blo != WXA_BOUNDARY_PEC) ||
            (bhi != WXA_BOUNDARY_PERIODIC && bhi != WXA_BOUNDARY_PEC) ||
            ((blo == WXA_BOUNDARY_PERIODIC) != (bhi == WXA_BOUNDARY_PERIODIC))) {
            delete s;
            return -2;   // a direction is periodic on both sides or on neither
        }
        s->pec_lo[d] = blo == WXA_BOUNDARY_PEC;
        s->pec_hi[d] = bhi == WXA_BOUNDARY_PEC;
        s->periodic[d] = blo == WXA_BOUNDARY_PERIODIC;
        s->any_pec = s->any_pec || s->pec_lo[d] || s->pec_hi[d];
        // boundary.particle_lo/hi: default = periodic with a periodic field boundary, absorbing otherwise;
        // periodic particles need a periodic field boundary and vice versa (WarpX::ReadBoundaryConditions)
        for (int side = 0; side < 2; ++side) {
            int32_t want = side == 0 ? cfg->particle_boundary_lo[d] : cfg->particle_boundary_hi[d];
            const bool fper = s->periodic[d] != 0;
            if (want == WXA_PBOUNDARY_DEFAULT) want = fper ? WXA_PBOUNDARY_PERIODIC : WXA_PBOUNDARY_ABSORBING;
            const bool ok = (want == WXA_PBOUNDARY_PERIODIC) == fper &&
                            (want == WXA_PBOUNDARY_PERIODIC || want == WXA_PBOUNDARY_ABSORBING ||
                             want == WXA_PBOUNDARY_REFLECTING);
            if (!ok) { delete s; return -2; }
            (side == 0 ? s->pbc_lo[d] : s->pbc_hi[d]) = want;
            s->any_particle_wall = s->any_particle_wall || want != WXA_PBOUNDARY_PERIODIC;
        }
    }
    // Yee staggering (Source/WarpX.cpp:2117-2125); warpx.grid_type = collocated: everything nodal (:2140-2152)
    // and the same shape factors in all directions for the gather (:967)
    int Es[3][3] = {{0, 1, 1}, {1, 0, 1}, {1, 1, 0}};
    int Bs[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (cfg->grid_type == WXA_GRID_COLLOCATED) {
        for (int c = 0; c < 3; ++c)
            for (int d = 0; d < 3; ++d) Es[c][d] = Bs[c][d] = 1;
        s->cfg.galerkin = 0;
    } else if (cfg->grid_type != WXA_GRID_STAGGERED) { delete s; return -2; }
    for (int c = 0; c < 3; ++c) {
        s->E[c].alloc(cfg->n_cell, Es[c], s->ng_EB);
        s->B[c].alloc(cfg->n_cell, Bs[c], s->ng_EB);
        s->J[c].alloc(cfg->n_cell, Es[c], s->ng_J);
        s->Ev[c] = s->E[c].v; s->Bv[c] = s->B[c].v; s->Jv[c] = s->J[c].v;
    }
    if (cfg->use_fdtd_nci_corr) {
        // WarpX::InitNCICorrector: cdtodz = c dt / dz, Galerkin tables unless the gather is nodal (:872-886)
        const double cdtodz = PhysConst::c * s->dt / s->dx[2];
        orc_nci_godfrey_stencil(cdtodz, s->cfg.galerkin ? 0 : 1, 0, s->nci_exeybz);
        orc_nci_godfrey_stencil(cdtodz, s->cfg.galerkin ? 0 : 1, 1, s->nci_bxbyez);
        for (int c = 0; c < 3; ++c) {
            s->nciE[c].alloc(cfg->n_cell, Es[c], s->ng_EB);
            s->nciB[c].alloc(cfg->n_cell, Bs[c], s->ng_EB);
        }
    }
    const int nodal[3] = {1, 1, 1};
    s->rho.alloc(cfg->n_cell, nodal, s->ng_rho);
    *out = s;
    return 0;
}

void orc_sim_destroy(orc_sim* s) { delete s; }

int orc_sim_add_species(orc_sim* s, double charge, double mass, const wxa_particle_view* init, int32_t* id) {
    auto sp = std::make_unique<Species>();
    sp->q = charge; sp->m = mass;
    const double* src[7] = {init->x, init->y, init->z, init->w, init->ux, init->uy, init->uz};
    for (int c = 0; c < 7; ++c) sp->a[c].assign(src[c], src[c] + init->np);
    if (init->idcpu) sp->id.assign(init->idcpu, init->idcpu + init->np);
    if (id) *id = (int32_t)s->species.size();
    s->species.push_back(std::move(sp));
    return 0;
}

// WarpX::Evolve (Source/Evolve/WarpXEvolve.cpp:94-347)
// BackTransformParticleFunctor (ComputeDiagFunctors/BackTransformParticleFunctor.cpp:76-152): SelectParticles (.H:49-62) and
// LorentzTransformParticles (.H:106-168) on host arrays; out[7][capacity], rows x y z w ux uy uz, in particle order
int orc_btd_select_particles(const wxa_particle_view* p, const double* const old6[6], double z_boost, double z_boost_old,
                             double t_boost, double dt, double t_lab, double gamma_boost, double* out, int64_t capacity,
                             int64_t* n_selected, void*) {
    if (!p || !old6 || !out || !n_selected || capacity <= 0 || !(gamma_boost > 1.0)) return -1;
    const double m_gammaboost = gamma_boost, m_betaboost = std::sqrt(1.0 - 1.0 / (gamma_boost * gamma_boost));
    const double m_Phys_c = PhysConst::c, m_inv_c2 = 1.0 / (m_Phys_c * m_Phys_c);
    const double m_uzfrm = -m_gammaboost * m_betaboost * m_Phys_c;
    int64_t n = 0;
    for (int64_t i = 0; i < p->np; ++i) {
        if (p->idcpu && p->idcpu[i] == WXA_IDCPU_RETIRED) continue;
        const double zp = p->z[i], zpold = old6[2][i];
        if (!(((zp >= z_boost) && (zpold <= z_boost_old)) || ((zp <= z_boost) && (zpold >= z_boost_old)))) continue;
        const double gamma_new_p = std::sqrt(1.0 + m_inv_c2 * (p->ux[i] * p->ux[i] + p->uy[i] * p->uy[i] + p->uz[i] * p->uz[i]));
        const double gamma_old_p = std::sqrt(1.0 + m_inv_c2 * (old6[3][i] * old6[3][i] + old6[4][i] * old6[4][i] + old6[5][i] * old6[5][i]));
        const double t_new_p = m_gammaboost * t_boost - m_uzfrm * zp * m_inv_c2;
        const double z_new_p = m_gammaboost * (zp + m_betaboost * m_Phys_c * t_boost);
        const double uz_new_p = m_gammaboost * p->uz[i] - gamma_new_p * m_uzfrm;
        const double t_old_p = m_gammaboost * (t_boost - dt) - m_uzfrm * zpold * m_inv_c2;
        const double z_old_p = m_gammaboost * (zpold + m_betaboost * m_Phys_c * (t_boost - dt));
        const double uz_old_p = m_gammaboost * old6[5][i] - gamma_old_p * m_uzfrm;
        const double weight_old = (t_new_p - t_lab) / (t_new_p - t_old_p);
        const double weight_new = (t_lab - t_old_p) / (t_new_p - t_old_p);
        if (n < capacity) {
            out[0 * capacity + n] = old6[0][i] * weight_old + p->x[i] * weight_new;
            out[1 * capacity + n] = old6[1][i] * weight_old + p->y[i] * weight_new;
            out[2 * capacity + n] = z_old_p * weight_old + z_new_p * weight_new;
            out[3 * capacity + n] = p->w[i];
            out[4 * capacity + n] = old6[3][i] * weight_old + p->ux[i] * weight_new;
            out[5 * capacity + n] = old6[4][i] * weight_old + p->uy[i] * weight_new;
            out[6 * capacity + n] = uz_old_p * weight_old + uz_new_p * weight_new;
        }
        ++n;
    }
    *n_selected = n;
    return 0;
}

int orc_sim_compute_rho(orc_sim* s);

// ---- back-transformed diagnostics, fields (Source/Diagnostics/BTDiagnostics.cpp, ComputeDiagFunctors/BackTransformFunctor.cpp)
// Lab-frame snapshot i is the set of events t_lab = t_i; at boosted time t it is the plane
//   z_boost = (t_i / gamma - t) c / beta     (BTDiagnostics.H:276-280),  lab position z_lab = (t_i - t / gamma) c / beta (:285-289)
// which sweeps down through the boosted domain; each step contributes one lab-frame slice, dz_lab = c dt / (beta gamma)
// apart (:885-890).  amrex::get_slice_data(interpolate = true) is restated as a linear interpolation between the two cell
// centres around z_boost (AMReX is not on disk; pinned since round 5 by the reference's test_3d_laser_acceleration_btd.json,
// which the host layer built on these routines meets at 3e-10).
static double btd_dz_lab(const orc_sim* s) { return PhysConst::c * s->dt * 1.0 / s->beta_boost * 1.0 / s->gamma_boost; }
static int btd_k_index(const orc_sim::BtdSnapshot& b, double dzl) {   // k_index_zlab (:892-905)
    return (int)std::floor((b.z_lab - b.zlo_lab) / dzl) + b.ksmall;
}
static bool btd_slice_in_domain(const orc_sim* s, const orc_sim::BtdSnapshot& b) {   // GetZSliceInDomainFlag (:999-1018)
    const double cs = s->dx[2];
    return !((b.z_boost <= s->plo[2] + 0.5 * cs) || (b.z_boost >= s->phi[2] - 0.5 * cs) || (b.z_lab <= b.zlo_lab) ||
             (b.z_lab >= b.zhi_lab));
}
// cell-centred value of a staggered component at cell (i, j, k): CellCenterFunctor -> ablastr::coarsen::Interp (sample.H:47-99)
static double btd_cell_value(const wxa_field_view& f, int i, int j, int k) {
    const Arr a(f);
    const int np[3] = {1 + f.stag[0], 1 + f.stag[1], 1 + f.stag[2]};
    const double wx = 1.0 / np[0], wy = 1.0 / np[1], wz = 1.0 / np[2];
    double c = 0.0;
    for (int kr = 0; kr < np[2]; ++kr)
        for (int jr = 0; jr < np[1]; ++jr)
            for (int ir = 0; ir < np[0]; ++ir) c += wx * wy * wz * a(i + ir, j + jr, k + kr);
    return c;
}
static void btd_compute_and_pack(orc_sim* s) {
    const double dzl = btd_dz_lab(s);
    const double g = s->gamma_boost, b = s->beta_boost, clight = PhysConst::c, inv_clight = 1.0 / PhysConst::c;
    bool any = false;
    for (auto& sn : s->btd) {   // PrepareBufferData (:755-776)
        sn.z_boost = (sn.t_lab / g - s->cur_time) * clight / b;
        sn.z_lab = (sn.t_lab - s->cur_time / g) * clight / b;
        any = any || (btd_slice_in_domain(s, sn) && !sn.full);
    }
    if (any) orc_sim_compute_rho(s);
    for (auto& sn : s->btd) {
        const bool in_domain = btd_slice_in_domain(s, sn);
        const int k_lab = btd_k_index(sn, dzl);
        if (in_domain && !sn.full && k_lab >= sn.ksmall && k_lab <= sn.kbig) {
            // BackTransformFunctor::operator() (:49-149): slice, Lorentz transform, copy to k_lab
            const double cs = s->dx[2];
            const int kc = (int)std::floor((sn.z_boost - s->plo[2]) / cs);
            const double zc = s->plo[2] + (kc + 0.5) * cs;
            int klo, khi;
            double w;
            if (sn.z_boost >= zc) { klo = kc; khi = kc + 1; w = (sn.z_boost - zc) / cs; }
            else { klo = kc - 1; khi = kc; w = (sn.z_boost - (zc - cs)) / cs; }
            if (klo >= 0 && khi < s->cfg.n_cell[2]) {
                const wxa_field_view* comp[10] = {&s->Ev[0], &s->Ev[1], &s->Ev[2], &s->Bv[0], &s->Bv[1], &s->Bv[2],
                                                  &s->Jv[0], &s->Jv[1], &s->Jv[2], &s->rho.v};
                const size_t plane = (size_t)sn.n[0] * sn.n[1], vol = plane * (size_t)sn.n[2];
                const size_t kk = (size_t)(k_lab - sn.ksmall);
                for (int j = 0; j < sn.n[1]; ++j)
                    for (int i = 0; i < sn.n[0]; ++i) {
                        double v[10];
                        for (int c = 0; c < 10; ++c)
                            v[c] = (1.0 - w) * btd_cell_value(*comp[c], i, j, klo) + w * btd_cell_value(*comp[c], i, j, khi);
                        // LorentzTransformZ (:246-317)
                        const double ex_lab = g * (v[0] + b * clight * v[4]);
                        const double by_lab = g * (v[4] + b * inv_clight * v[0]);
                        v[0] = ex_lab; v[4] = by_lab;
                        const double ey_lab = g * (v[1] - b * clight * v[3]);
                        const double bx_lab = g * (v[3] - b * inv_clight * v[1]);
                        v[1] = ey_lab; v[3] = bx_lab;
                        const double j_lab = g * (v[8] + b * clight * v[9]);
                        const double rho_lab = g * (v[9] + b * inv_clight * v[8]);
                        v[8] = j_lab; v[9] = rho_lab;
                        for (int c = 0; c < 10; ++c) sn.data[(size_t)c * vol + kk * plane + (size_t)j * sn.n[0] + i] = v[c];
                    }
            }
        }
        // UpdateBufferData (:778-798), and the flags DoDump / Flush leave behind (:294-320, :907-914)
        if (in_domain) ++sn.counter;
        if (k_lab == sn.ksmall) sn.last_valid = 1;
        if (sn.last_valid == 1) sn.full = 1;
    }
}

// BTDiagnostics::ReadParameters (:206-292) + DerivedInitData (:66-205) + InitializeBufferData (:333-506), one level, one box,
// diag domain = the whole boosted-frame domain
int orc_sim_add_btd(orc_sim* s, int32_t num_snapshots, double dt_snapshots_lab, int32_t buffer_size, int32_t write_species) {
    if (!s || num_snapshots < 1 || !(dt_snapshots_lab > 0.0) || buffer_size < 1) return -1;
    if (!(s->gamma_boost > 1.0) || s->mw_dir != 2) return -1;
    s->btd_write_species = write_species != 0;
    s->btd_buffer_size = buffer_size;
    s->btd_dt_snap = dt_snapshots_lab;
    const double g = s->gamma_boost, b = s->beta_boost;
    const double mw_beta = s->mw_on ? s->mw_v / PhysConst::c : 0.0;
    const double boosted_mw_v = (mw_beta - b) / (1.0 - b * mw_beta);
    const double dzl = btd_dz_lab(s);
    s->btd.assign((size_t)num_snapshots, orc_sim::BtdSnapshot{});
    for (int i = 0; i < num_snapshots; ++i) {
        auto& sn = s->btd[(size_t)i];
        sn.t_lab = i * dt_snapshots_lab + g * b * s->phi[2] / PhysConst::c;
        double dlo[3], dhi[3];
        for (int d = 0; d < 3; ++d) {
            int lo = std::max(0, (int)std::floor((s->plo[d] - s->plo[d]) / s->dx[d]));
            int hi = std::max(0, (int)std::ceil((s->phi[d] - s->plo[d]) / s->dx[d])) - 1;
            if (hi <= lo) hi = lo + 1;
            dlo[d] = s->plo[d] + lo * s->dx[d];
            dhi[d] = s->plo[d] + (hi + 1) * s->dx[d];
        }
        const double zmin_lab = (dlo[2] - boosted_mw_v * s->cur_time) * (1.0 - b * mw_beta) * g;
        const double zmax_lab = (dhi[2] - boosted_mw_v * s->cur_time) * (1.0 - b * mw_beta) * g;
        sn.z_boost = (sn.t_lab / g - s->cur_time) * PhysConst::c / b;
        sn.z_lab = (sn.t_lab - s->cur_time / g) * PhysConst::c / b;
        const int nz_lab = std::max(0, (int)std::floor((zmax_lab - zmin_lab) / dzl));
        sn.n[0] = std::max(0, (int)std::floor((dhi[0] - dlo[0]) / s->dx[0]));
        sn.n[1] = std::max(0, (int)std::floor((dhi[1] - dlo[1]) / s->dx[1]));
        const int nzs = (int)std::ceil((double)nz_lab / (double)buffer_size) * buffer_size;
        sn.n[2] = nzs;
        sn.zlo_lab = zmin_lab + s->mw_v * sn.t_lab;
        sn.zhi_lab = zmax_lab + s->mw_v * sn.t_lab;
        sn.zhi_lab = sn.zhi_lab + 0.5 * dzl;
        sn.zlo_lab = sn.zhi_lab - nzs * dzl;
        sn.kbig = (int)std::floor((sn.zhi_lab - (sn.zlo_lab + 0.5 * dzl)) / dzl);
        sn.ksmall = sn.kbig - (nzs - 1);
        sn.data.assign((size_t)10 * (size_t)nzs * (size_t)sn.n[1] * (size_t)sn.n[0], 0.0);
    }
    return 0;
}
int orc_sim_btd_info(orc_sim* s, int32_t i, int32_t n[3], double z_lab[2], double* t_lab, int32_t* filled, int32_t* full) {
    if (!s || i < 0 || i >= (int32_t)s->btd.size()) return -1;
    const auto& sn = s->btd[(size_t)i];
    if (n) for (int d = 0; d < 3; ++d) n[d] = sn.n[d];
    if (z_lab) { z_lab[0] = sn.zlo_lab; z_lab[1] = sn.zhi_lab; }
    if (t_lab) *t_lab = sn.t_lab;
    if (filled) *filled = sn.counter;
    if (full) *full = sn.full;
    return 0;
}
int orc_sim_btd_num_particles(orc_sim* s, int32_t i, int32_t id, int64_t* n) {
    if (!s || !n || i < 0 || i >= (int32_t)s->btd.size() || id < 0) return -1;
    const auto& sn = s->btd[(size_t)i];
    *n = id < (int32_t)sn.particles.size() ? (int64_t)sn.particles[(size_t)id][0].size() : 0;
    return 0;
}
int orc_sim_btd_particles(orc_sim* s, int32_t i, int32_t id, double* out) {
    int64_t n = 0;
    if (orc_sim_btd_num_particles(s, i, id, &n) != 0 || !out) return -1;
    if (n == 0) return 0;
    const auto& rows = s->btd[(size_t)i].particles[(size_t)id];
    for (int c = 0; c < 7; ++c) std::memcpy(out + (size_t)c * (size_t)n, rows[(size_t)c].data(), sizeof(double) * (size_t)n);
    return 0;
}
int orc_sim_btd_data(orc_sim* s, int32_t i, int32_t comp, double* out) {
    if (!s || !out || i < 0 || i >= (int32_t)s->btd.size() || comp < 0 || comp >= 10) return -1;
    const auto& sn = s->btd[(size_t)i];
    const size_t n = (size_t)sn.n[0] * sn.n[1] * sn.n[2];
    std::memcpy(out, sn.data.data() + (size_t)comp * n, sizeof(double) * n);
    return 0;
}

// Reduced diagnostics of the oracle stepper (Source/Diagnostics/ReducedDiags/): the row of one diagnostic from the
// formulas above (orc_field_energy, orc_particle_energy, orc_particle_momentum), laid out as FieldEnergy.cpp:146-150,
// ParticleEnergy.cpp:160-210, ParticleMomentum.cpp:175-240 and ParticleNumber.cpp:97-127 lay out m_data.
static void rd_compute(orc_sim* s, orc_sim::ReducedDiag& rd) {
    const int ns = (int)s->species.size();
    constexpr double tiny = std::numeric_limits<double>::min();
    if (rd.type == "FieldEnergy") {
        double out[3];
        orc_field_energy(s->Ev, s->Bv, s->dx, out);
        rd.data.assign(out, out + 3);
    } else if (rd.type == "ParticleEnergy") {
        rd.data.assign((size_t)(2 * ns + 2), 0.0);
        double Wtot = 0.0;
        for (int i = 0; i < ns; ++i) {
            wxa_particle_view p = s->species[i]->view();
            const double E = orc_particle_energy(&p, s->species[i]->m);
            long double W = 0.0L;
            for (int64_t k = 0; k < p.np; ++k) W += p.w[k];
            rd.data[1 + i] = E;
            rd.data[1 + ns + 1 + i] = (double)W > tiny ? E / (double)W : 0.0;
            rd.data[0] += E;
            Wtot += (double)W;
        }
        rd.data[1 + ns] = Wtot > tiny ? rd.data[0] / Wtot : 0.0;
    } else if (rd.type == "ParticleMomentum") {
        rd.data.assign((size_t)(6 * ns + 6), 0.0);
        double Wtot = 0.0;
        for (int i = 0; i < ns; ++i) {
            wxa_particle_view p = s->species[i]->view();
            double P[3];
            orc_particle_momentum(&p, s->species[i]->m, P);
            long double W = 0.0L;
            for (int64_t k = 0; k < p.np; ++k) W += p.w[k];
            for (int d = 0; d < 3; ++d) {
                rd.data[3 + 3 * i + d] = P[d];
                rd.data[3 + 3 * ns + 3 + 3 * i + d] = (double)W > tiny ? P[d] / (double)W : 0.0;
                rd.data[d] += P[d];
            }
            Wtot += (double)W;
        }
        for (int d = 0; d < 3; ++d) rd.data[3 + 3 * ns + d] = Wtot > tiny ? rd.data[d] / Wtot : 0.0;
    } else {   // ParticleNumber
        rd.data.assign((size_t)(2 * ns + 2), 0.0);
        for (int i = 0; i < ns; ++i) {
            wxa_particle_view p = s->species[i]->view();
            long double W = 0.0L;
            int64_t live = 0;
            for (int64_t k = 0; k < p.np; ++k) {
                if (p.idcpu && p.idcpu[k] == WXA_IDCPU_RETIRED) continue;
                W += p.w[k];
                ++live;
            }
            rd.data[1 + i] = (double)live;
            rd.data[1 + ns + 1 + i] = (double)W;
            rd.data[0] += (double)live;
            rd.data[1 + ns] += (double)W;
        }
    }
}

// ComputeDiags(step) + WriteToFile(step) (WarpXEvolve.cpp:299-305; ReducedDiags.cpp:97-125); step = -1 before the first step
static void rd_compute_and_write(orc_sim* s, int step) {
    for (auto& rd : s->rdiags) {
        const int n = step + 1;
        if (!(rd.period > 0 && (n - rd.start) % rd.period == 0 && n >= rd.start && n <= rd.stop)) continue;
        rd_compute(s, rd);
        if (rd.file.empty()) continue;
        FILE* f = std::fopen(rd.file.c_str(), "a");
        if (!f) continue;
        std::fprintf(f, "%d %.14e", n, s->cur_time);
        for (double v : rd.data) std::fprintf(f, " %.14e", v);
        std::fprintf(f, "\n");
        std::fclose(f);
    }
}

// one slice only ("period" or "start:stop:period"): what the oracle's own tests use
int orc_sim_add_reduced_diag(orc_sim* s, const char* name, const char* type, const char* intervals, const char* path) {
    if (!s || !name || !type) return -1;
    const std::string t = type;
    if (t != "FieldEnergy" && t != "ParticleEnergy" && t != "ParticleMomentum" && t != "ParticleNumber") return -2;
    orc_sim::ReducedDiag rd;
    rd.name = name;
    rd.type = t;
    if (intervals && *intervals) {
        int a = 0, b = 0, c = 0;
        if (std::strchr(intervals, ',')) return -3;
        if (std::sscanf(intervals, "%d:%d:%d", &a, &b, &c) == 3) { rd.start = a; rd.stop = b; rd.period = c; }
        else if (!std::strchr(intervals, ':') && std::sscanf(intervals, "%d", &a) == 1) rd.period = a;
        else return -3;
    }
    if (path && *path) {   // the directory must exist (the host layer creates it; the oracle is a checker)
        rd.file = std::string(path) + rd.name + ".txt";
        FILE* f = std::fopen(rd.file.c_str(), "w");
        if (!f) return -4;
        std::fclose(f);   // rows only: the header row is the host layer's business
    }
    s->rdiags.push_back(std::move(rd));
    return 0;
}

int orc_sim_reduced_diag_data(orc_sim* s, const char* name, int32_t compute_now, double* out, int32_t capacity, int32_t* n) {
    if (!s || !name || !n) return -1;
    for (auto& rd : s->rdiags)
        if (rd.name == name) {
            if (compute_now) rd_compute(s, rd);
            *n = (int32_t)rd.data.size();
            for (int32_t i = 0; i < std::min(*n, capacity); ++i) out[i] = rd.data[(size_t)i];
            return 0;
        }
    return -2;
}

int orc_sim_evolve(orc_sim* s, int32_t numsteps) {
    if (!s->rdiags_started) {   // WarpXInitData.cpp:612-619
        s->rdiags_started = true;
        rd_compute_and_write(s, (int)s->istep - 1);
    }
    for (int32_t step = 0; step < numsteps; ++step) {
        // ExplicitFillBoundaryEBUpdateAux (:473-531)
        if (s->is_synchronized) {
            fill_boundary_EB(s, s->Ev, s->ng_EB, false);
            fill_boundary_EB(s, s->Bv, s->ng_EB, false);
            push_p_all(s, -0.5 * s->dt);
            s->is_synchronized = false;
        } else {
            fill_boundary_EB(s, s->Ev, s->ng_gather, false);
            fill_boundary_EB(s, s->Bv, s->ng_gather, false);
        }
        one_step_nosub(s);
        if (step == numsteps - 1) {
            // Synchronize (:65-93)
            fill_boundary_EB(s, s->Ev, s->ng_gather, false);
            fill_boundary_EB(s, s->Bv, s->ng_gather, false);
            push_p_all(s, 0.5 * s->dt);
            s->is_synchronized = true;
        }
        s->istep++;
        s->cur_time += s->dt;
        // WarpXEvolve.cpp:241 multi_diags->FilterComputePackFlush(step, false, true): the BackTransformed diagnostics run
        // before the window moves and before the particles meet the boundaries (MultiDiagnostics.cpp:81-96)
        if (!s->btd.empty()) btd_compute_and_pack(s);
        move_window(s, /*move_j=*/s->is_synchronized);   // :246 MoveWindow(step+1, move_j)
        for (auto& L : s->lasers) {
            // lasers are particle containers too (MultiParticleContainer::ApplyBoundaryConditions loops over all of them,
            // MultiParticleContainer.cpp:629-635): in a boosted frame the antenna drifts with -beta c and leaves through
            // the lower wall, where its particles are absorbed and dropped like any others; then the periodic wrap
            Species& sp = L->parts;
            if (s->any_particle_wall && !sp.a[0].empty()) {
                if (sp.id.empty()) sp.id.assign(sp.a[0].size(), 0);
                wxa_particle_view p = sp.view();
                int64_t lost = 0;
                orc_apply_particle_boundaries(&p, s->plo, s->phi, s->pbc_lo, s->pbc_hi, &lost, nullptr, nullptr);
                if (lost > 0) {
                    size_t keep = 0;
                    for (size_t ip = 0; ip < sp.id.size(); ++ip) {
                        if (sp.id[ip] == WXA_IDCPU_RETIRED) continue;
                        for (int c = 0; c < 7; ++c) sp.a[c][keep] = sp.a[c][ip];
                        sp.id[keep] = sp.id[ip];
                        ++keep;
                    }
                    for (int c = 0; c < 7; ++c) sp.a[c].resize(keep);
                    sp.id.resize(keep);
                }
            }
            wxa_particle_view p = sp.view();
            orc_enforce_periodic(&p, s->plo, s->phi, s->periodic, nullptr);
        }
        {   // HandleParticlesAtBoundaries (:533-581): ApplyBoundaryConditions, then the periodic wrap of Redistribute
            Tic t(s, 6);
            for (auto& sp : s->species) {
                if (s->any_particle_wall) {
                    if (sp->id.empty()) sp->id.assign(sp->a[0].size(), 0);
                    wxa_particle_view p = sp->view();
                    int64_t lost = 0;
                    orc_apply_particle_boundaries(&p, s->plo, s->phi, s->pbc_lo, s->pbc_hi, &lost,
                                                  nullptr, nullptr);
                    if (lost > 0) {   // Redistribute drops the invalidated particles
                        size_t keep = 0;
                        for (size_t ip = 0; ip < sp->id.size(); ++ip) {
                            if (sp->id[ip] == WXA_IDCPU_RETIRED) continue;
                            for (int c = 0; c < 7; ++c) sp->a[c][keep] = sp->a[c][ip];
                            sp->id[keep] = sp->id[ip];
                            ++keep;
                        }
                        for (int c = 0; c < 7; ++c) sp->a[c].resize(keep);
                        sp->id.resize(keep);
                    }
                }
                wxa_particle_view p = sp->view();
                orc_enforce_periodic(&p, s->plo, s->phi, s->periodic, nullptr);
            }
        }
        rd_compute_and_write(s, (int)s->istep - 1);     // WarpXEvolve.cpp:299-305
    }
    return 0;
}

// warpx.do_moving_window (Source/WarpX.cpp:640-660): the window starts at prob_lo
int orc_sim_set_moving_window(orc_sim* s, const wxa_moving_window* mw) {
    if (!s || !mw || mw->dir < 0 || mw->dir > 2 || !(mw->v > 0.0)) return -1;
    if (s->periodic[mw->dir]) return -2;   // the window direction cannot be periodic
    s->mw_on = true;
    s->mw_dir = mw->dir;
    s->mw_v = mw->v * PhysConst::c;
    s->mw_x = s->plo[mw->dir];
    for (auto& sp : s->species) sp->inj_pos = s->phi[mw->dir];   // Source/WarpX.cpp:301
    return 0;
}

// <species>.injection_style = NUniformPerCell, profile = constant, momentum at_rest (+ continuous injection):
// add_initial != 0 fills the current domain now (PhysicalParticleContainer::InitData -> AddParticles)
int orc_sim_set_injection(orc_sim* s, int32_t id, const wxa_plasma_injector* inj, int add_initial, int continuous) {
    if (!s || !inj || id < 0 || id >= (int32_t)s->species.size()) return -1;
    Species& sp = *s->species[id];
    sp.inj = *inj;
    sp.inject = continuous != 0;
    sp.inj_pos = s->mw_on ? s->phi[s->mw_dir] : 0.0;
    if (add_initial) add_plasma(s, sp, s->plo, s->phi);
    return 0;
}

int orc_sim_set_external_particle_fields(orc_sim* s, int32_t id, const double E[3], const double B[3]) {
    if (!s || !E || !B || id < 0 || id >= (int32_t)s->species.size()) return -1;
    for (int d = 0; d < 3; ++d) { s->species[id]->ext_eb[d] = E[d]; s->species[id]->ext_eb[3 + d] = B[d]; }
    return 0;
}

int orc_sim_set_radiation_reaction(orc_sim* s, int32_t id, int32_t on) {
    if (!s || id < 0 || id >= (int32_t)s->species.size()) return -1;
    s->species[id]->do_crr = on != 0;
    return 0;
}

int orc_sim_add_laser(orc_sim* s, const wxa_laser_antenna* la) {
    if (!s || !la || !(la->e_max > 0) || !(la->wavelength > 0)) return -1;
    auto L = std::make_unique<LaserAntenna>();
    L->cfg = *la;
    L->parts.q = 1.0; L->parts.m = 1e300;
    double n = 0, p = 0;
    for (int d = 0; d < 3; ++d) { n += la->direction[d] * la->direction[d]; p += la->polarization[d] * la->polarization[d]; }
    const double sn = 1.0 / std::sqrt(n), spx = 1.0 / std::sqrt(p);
    for (int d = 0; d < 3; ++d) { L->nvec[d] = la->direction[d] * sn; L->p_X[d] = la->polarization[d] * spx; L->position[d] = la->position[d]; }
    double dp = 0;
    for (int d = 0; d < 3; ++d) dp += L->nvec[d] * L->p_X[d];
    if (std::abs(dp) >= 1e-14) return -2;   // polarization must lie in the antenna plane
    if (s->gamma_boost > 1.0) {   // :182-197 the antenna sits at Z0_lab / gamma_boost; the boost must be along the laser
        if (L->nvec[0] * L->nvec[0] + L->nvec[1] * L->nvec[1] + (L->nvec[2] - 1.0) * (L->nvec[2] - 1.0) >= 1.e-12) return -2;
        L->Z0_lab = L->nvec[0] * L->position[0] + L->nvec[1] * L->position[1] + L->nvec[2] * L->position[2];
        const double Z0_boost = L->Z0_lab / s->gamma_boost;
        for (int d = 0; d < 3; ++d) L->position[d] += (Z0_boost - L->Z0_lab) * L->nvec[d];
    }
    // p_Y = nvec x p_X (:222)
    L->p_Y[0] = L->nvec[1] * L->p_X[2] - L->nvec[2] * L->p_X[1];
    L->p_Y[1] = L->nvec[2] * L->p_X[0] - L->nvec[0] * L->p_X[2];
    L->p_Y[2] = L->nvec[0] * L->p_X[1] - L->nvec[1] * L->p_X[0];
    laser_init(s, *L);
    s->lasers.push_back(std::move(L));
    return 0;
}

double orc_sim_dt(const orc_sim* s) { return s->dt; }
int64_t orc_sim_istep(const orc_sim* s) { return s->istep; }

int orc_sim_get_field(orc_sim* s, const char* name, wxa_field_view* out) {
    const std::string n(name);
    const char* comps = "xyz";
    for (int c = 0; c < 3; ++c) {
        if (n == std::string("E") + comps[c]) { *out = s->Ev[c]; return 0; }
        if (n == std::string("B") + comps[c]) { *out = s->Bv[c]; return 0; }
        if (n == std::string("j") + comps[c]) { *out = s->Jv[c]; return 0; }
    }
    if (n == "rho") { *out = s->rho.v; return 0; }
    return -1;
}

int orc_sim_get_particles(orc_sim* s, int32_t id, wxa_particle_view* out) {
    if (id < 0 || id >= (int32_t)s->species.size()) return -1;
    *out = s->species[id]->view();
    return 0;
}

// RhoFunctor (Source/Diagnostics/ComputeDiagFunctors/RhoFunctor.cpp:42-61):
// GetChargeDensity (all species) + ApplyFilterandSumBoundaryRho
int32_t orc_sim_halo_overlap(const orc_sim*) { return 0; }   // one brick: nothing to overlap

int orc_sim_compute_rho(orc_sim* s) {
    orc_field_set_zero(&s->rho.v, nullptr);
    const wxa_grid_geom g = s->geom_for(s->ng_rho);
    for (auto& sp : s->species) {
        wxa_particle_view p = sp->view();
        orc_deposit_charge(&p, &s->rho.v, &g, sp->q, s->cfg.nox, nullptr);
    }
    for (auto& L : s->lasers) {
        wxa_particle_view p = L->parts.view();
        orc_deposit_charge(&p, &s->rho.v, &g, 1.0, s->cfg.nox, nullptr);
    }
    // WarpXParticleContainer::DepositCharge (:1285-1290): each species' rho is reflected over the PEC walls
    // right after its deposition (linear: done once on the total), before the filter and the sum
    // (the stepper follows the reference: the valid points only, PEC::ApplyReflectiveBoundarytoRhofield :697-698 -- the
    // guard columns of the wall-free directions keep their deposits, see orc_apply_pec_rho)
    if (s->any_pec) {
        const bool tangent[3] = {true, true, true};
        reflect_over_pec(s->rho.v, tangent, s->dom_lo, s->dom_hi, s->pec_lo, s->pec_hi, /*transverse_guards=*/false);
    }
    if (s->cfg.use_filter) {
        Field tmp; tmp.v = s->rho.v; tmp.data.assign(s->rho.data.size(), 0.0); tmp.v.p = tmp.data.data();
        orc_filter_bilinear(&s->rho.v, &tmp.v, nullptr);
        s->rho.data = tmp.data; s->rho.v.p = s->rho.data.data();
    }
    orc_sum_boundary_periodic(&s->rho.v, s->ng_rho, s->periodic, nullptr);
    return 0;
}

int orc_sim_get_timers(orc_sim* s, double ms[8], int64_t counts[8], int reset) {
    for (int i = 0; i < 8; ++i) { ms[i] = s->timers[i]; counts[i] = s->counts[i]; }
    if (reset) for (int i = 0; i < 8; ++i) { s->timers[i] = 0; s->counts[i] = 0; }
    return 0;
}
int orc_sim_enable_timers(orc_sim* s, int e) { s->do_timers = e != 0; return 0; }

}  // extern "C"
