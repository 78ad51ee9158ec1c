// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (C++17, fp64) of the arithmetic of WarpX's explicit-FDTD PIC
// inner loop, following the reference headers operation by operation.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
// and only as the checker.  Paths cited are relative to the WarpX checkout
// @ 2024-10-24 (/root/reference).
//
// Parity pin: the step-level driver in pic_oracle.cpp reproduces, at the
// reference's own tolerance (rtol 1e-9), the golden checksums under
// Regression/Checksum/benchmarks_json/ of
//   test_3d_langmuir_multi        (Esirkepov order 1, Boris, Yee; tests/test_oracle_golden.py)
//   test_3d_langmuir_multi_picmi  (direct deposition on the Yee grid, filter, 8 ppc, gather without
//                                  Galerkin shapes; same file), test_3d_langmuir_multi_nodal (collocated),
//   test_3d_pec_field, test_3d_pec_particle (order 3, Vay, filter, PEC), test_3d_particle_boundaries,
//   test_3d_laser_acceleration    (moving window, injection, antenna; tests/test_pec_golden.py).
// The per-kernel outputs are not pinned by any reference test (the reference has
// no unit tests).
//
// AMReX itself (containers, FillBoundary, SumBoundary, Redistribute) is not in
// the reference tree (cmake/dependencies/AMReX.cmake:283-288 fetches commit
// 62c2a81eac7862d526e5861ef2befc00b7f5b759); its semantics are restated from
// the WarpX call sites and docs.
#ifndef ORACLE_PIC_KERNELS_HPP_
#define ORACLE_PIC_KERNELS_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../include/warpx_amd.h"

namespace orc {

// Source/ablastr/constant.H:41-50 (CODATA 2018), digit for digit.
namespace PhysConst {
constexpr double c = 299'792'458.;
constexpr double ep0 = 8.8541878128e-12;
constexpr double mu0 = 1.25663706212e-06;
constexpr double q_e = 1.602176634e-19;
constexpr double m_e = 9.1093837015e-31;
constexpr double r_e = 2.817940326204929e-15;   // classical electron radius (constant.H:63)
constexpr double m_p = 1.67262192369e-27;
}

// amrex::Array4 accessor on a wxa_field_view (Fortran order, guards included).
struct Arr {
    double* p; int lo0, lo1, lo2; int64_t js, ks;
    explicit Arr(const wxa_field_view& v)
        : p(v.p), lo0(v.lo[0]), lo1(v.lo[1]), lo2(v.lo[2]), js(v.jstride), ks(v.kstride) {}
    inline double& operator()(int i, int j, int k) const {
        return p[(i - lo0) + (j - lo1) * js + (k - lo2) * ks];
    }
};

constexpr int NODE = 1;  // amrex::IndexType::NODE
constexpr int CELL = 0;  // amrex::IndexType::CELL

// ---------------------------------------------------------------------------
// Source/Particles/ShapeFactors.H:27-84
template <int depos_order>
inline int compute_shape_factor(double* sx, double xmid) {
    if constexpr (depos_order == 0) {
        const auto j = static_cast<int>(xmid + 0.5);
        sx[0] = 1.0;
        return j;
    } else if constexpr (depos_order == 1) {
        const auto j = static_cast<int>(xmid);
        const double xint = xmid - double(j);
        sx[0] = 1.0 - xint;
        sx[1] = xint;
        return j;
    } else if constexpr (depos_order == 2) {
        const auto j = static_cast<int>(xmid + 0.5);
        const double xint = xmid - double(j);
        sx[0] = 0.5 * (0.5 - xint) * (0.5 - xint);
        sx[1] = 0.75 - xint * xint;
        sx[2] = 0.5 * (0.5 + xint) * (0.5 + xint);
        return j - 1;
    } else if constexpr (depos_order == 3) {
        const auto j = static_cast<int>(xmid);
        const double xint = xmid - double(j);
        sx[0] = (1.0) / (6.0) * (1.0 - xint) * (1.0 - xint) * (1.0 - xint);
        sx[1] = (2.0) / (3.0) - xint * xint * (1.0 - xint / (2.0));
        sx[2] = (2.0) / (3.0) - (1.0 - xint) * (1.0 - xint) * (1.0 - 0.5 * (1.0 - xint));
        sx[3] = (1.0) / (6.0) * xint * xint * xint;
        return j - 1;
    } else if constexpr (depos_order == 4) {   // :67-77
        const auto j = static_cast<int>(xmid + 0.5);
        const double xint = xmid - double(j);
        sx[0] = (1.0) / (24.0) * (0.5 - xint) * (0.5 - xint) * (0.5 - xint) * (0.5 - xint);
        sx[1] = (1.0) / (24.0) * (4.75 - 11.0 * xint + 4.0 * xint * xint * (1.5 + xint - xint * xint));
        sx[2] = (1.0) / (24.0) * (14.375 + 6.0 * xint * xint * (xint * xint - 2.5));
        sx[3] = (1.0) / (24.0) * (4.75 + 11.0 * xint + 4.0 * xint * xint * (1.5 - xint - xint * xint));
        sx[4] = (1.0) / (24.0) * (0.5 + xint) * (0.5 + xint) * (0.5 + xint) * (0.5 + xint);
        return j - 2;
    } else {
        static_assert(depos_order <= 4, "orders 0..4");
        return 0;
    }
}

// Source/Particles/ShapeFactors.H:93-156
template <int depos_order>
inline int compute_shifted_shape_factor(double* sx, const double x_old, const int i_new) {
    if constexpr (depos_order == 1) {
        const auto i = static_cast<int>(std::floor(x_old));
        const int i_shift = i - i_new;
        const double xint = x_old - double(i);
        sx[1 + i_shift] = 1.0 - xint;
        sx[2 + i_shift] = xint;
        return i;
    } else if constexpr (depos_order == 2) {
        const auto i = static_cast<int>(x_old + 0.5);
        const int i_shift = i - (i_new + 1);
        const double xint = x_old - double(i);
        sx[1 + i_shift] = 0.5 * (0.5 - xint) * (0.5 - xint);
        sx[2 + i_shift] = 0.75 - xint * xint;
        sx[3 + i_shift] = 0.5 * (0.5 + xint) * (0.5 + xint);
        return i - 1;
    } else if constexpr (depos_order == 3) {
        const auto i = static_cast<int>(x_old);
        const int i_shift = i - (i_new + 1);
        const double xint = x_old - double(i);
        sx[1 + i_shift] = (1.0) / (6.0) * (1.0 - xint) * (1.0 - xint) * (1.0 - xint);
        sx[2 + i_shift] = (2.0) / (3.0) - xint * xint * (1.0 - xint / (2.0));
        sx[3 + i_shift] = (2.0) / (3.0) - (1.0 - xint) * (1.0 - xint) * (1.0 - 0.5 * (1.0 - xint));
        sx[4 + i_shift] = (1.0) / (6.0) * xint * xint * xint;
        return i - 1;
    } else if constexpr (depos_order == 4) {   // :138-149
        const auto i = static_cast<int>(x_old + 0.5);
        const int i_shift = i - (i_new + 2);
        const double xint = x_old - double(i);
        sx[1 + i_shift] = (1.0) / (24.0) * (0.5 - xint) * (0.5 - xint) * (0.5 - xint) * (0.5 - xint);
        sx[2 + i_shift] = (1.0) / (24.0) * (4.75 - 11.0 * xint + 4.0 * xint * xint * (1.5 + xint - xint * xint));
        sx[3 + i_shift] = (1.0) / (24.0) * (14.375 + 6.0 * xint * xint * (xint * xint - 2.5));
        sx[4 + i_shift] = (1.0) / (24.0) * (4.75 + 11.0 * xint + 4.0 * xint * xint * (1.5 - xint - xint * xint));
        sx[5 + i_shift] = (1.0) / (24.0) * (0.5 + xint) * (0.5 + xint) * (0.5 + xint) * (0.5 + xint);
        return i - 2;
    } else {
        static_assert(depos_order >= 1 && depos_order <= 4, "orders 1..4");
        return 0;
    }
}

// ---------------------------------------------------------------------------
// Source/Particles/Gather/FieldGather.H:36-194 (shape set-up) and :368-423 (3-D loops)
template <int depos_order, int galerkin_interpolation>
inline void doGatherShapeN(const double xp, const double yp, const double zp,
                           double& Exp, double& Eyp, double& Ezp,
                           double& Bxp, double& Byp, double& Bzp,
                           const Arr& ex_arr, const Arr& ey_arr, const Arr& ez_arr,
                           const Arr& bx_arr, const Arr& by_arr, const Arr& bz_arr,
                           const int* ex_type, const int* ey_type, const int* ez_type,
                           const int* bx_type, const int* by_type, const int* bz_type,
                           const double* dinv, const double* xyzmin, const int* lo) {
    constexpr int zdir = 2;
    constexpr int NG = depos_order + 1 - galerkin_interpolation;

    // x direction
    const double x = (xp - xyzmin[0]) * dinv[0];
    double sx_node[depos_order + 1];
    double sx_cell[depos_order + 1];
    double sx_node_galerkin[NG] = {0.};
    double sx_cell_galerkin[NG] = {0.};
    int j_node = 0, j_cell = 0, j_node_v = 0, j_cell_v = 0;
    if ((ey_type[0] == NODE) || (ez_type[0] == NODE) || (bx_type[0] == NODE)) {
        j_node = compute_shape_factor<depos_order>(sx_node, x);
    }
    if ((ey_type[0] == CELL) || (ez_type[0] == CELL) || (bx_type[0] == CELL)) {
        j_cell = compute_shape_factor<depos_order>(sx_cell, x - 0.5);
    }
    if ((ex_type[0] == NODE) || (by_type[0] == NODE) || (bz_type[0] == NODE)) {
        j_node_v = compute_shape_factor<depos_order - galerkin_interpolation>(sx_node_galerkin, x);
    }
    if ((ex_type[0] == CELL) || (by_type[0] == CELL) || (bz_type[0] == CELL)) {
        j_cell_v = compute_shape_factor<depos_order - galerkin_interpolation>(sx_cell_galerkin, x - 0.5);
    }
    const double* sx_ex = (ex_type[0] == NODE) ? sx_node_galerkin : sx_cell_galerkin;
    const double* sx_ey = (ey_type[0] == NODE) ? sx_node : sx_cell;
    const double* sx_ez = (ez_type[0] == NODE) ? sx_node : sx_cell;
    const double* sx_bx = (bx_type[0] == NODE) ? sx_node : sx_cell;
    const double* sx_by = (by_type[0] == NODE) ? sx_node_galerkin : sx_cell_galerkin;
    const double* sx_bz = (bz_type[0] == NODE) ? sx_node_galerkin : sx_cell_galerkin;
    int const j_ex = (ex_type[0] == NODE) ? j_node_v : j_cell_v;
    int const j_ey = (ey_type[0] == NODE) ? j_node : j_cell;
    int const j_ez = (ez_type[0] == NODE) ? j_node : j_cell;
    int const j_bx = (bx_type[0] == NODE) ? j_node : j_cell;
    int const j_by = (by_type[0] == NODE) ? j_node_v : j_cell_v;
    int const j_bz = (bz_type[0] == NODE) ? j_node_v : j_cell_v;

    // y direction
    const double y = (yp - xyzmin[1]) * dinv[1];
    double sy_node[depos_order + 1];
    double sy_cell[depos_order + 1];
    double sy_node_v[NG];
    double sy_cell_v[NG];
    int k_node = 0, k_cell = 0, k_node_v = 0, k_cell_v = 0;
    if ((ex_type[1] == NODE) || (ez_type[1] == NODE) || (by_type[1] == NODE)) {
        k_node = compute_shape_factor<depos_order>(sy_node, y);
    }
    if ((ex_type[1] == CELL) || (ez_type[1] == CELL) || (by_type[1] == CELL)) {
        k_cell = compute_shape_factor<depos_order>(sy_cell, y - 0.5);
    }
    if ((ey_type[1] == NODE) || (bx_type[1] == NODE) || (bz_type[1] == NODE)) {
        k_node_v = compute_shape_factor<depos_order - galerkin_interpolation>(sy_node_v, y);
    }
    if ((ey_type[1] == CELL) || (bx_type[1] == CELL) || (bz_type[1] == CELL)) {
        k_cell_v = compute_shape_factor<depos_order - galerkin_interpolation>(sy_cell_v, y - 0.5);
    }
    const double* sy_ex = (ex_type[1] == NODE) ? sy_node : sy_cell;
    const double* sy_ey = (ey_type[1] == NODE) ? sy_node_v : sy_cell_v;
    const double* sy_ez = (ez_type[1] == NODE) ? sy_node : sy_cell;
    const double* sy_bx = (bx_type[1] == NODE) ? sy_node_v : sy_cell_v;
    const double* sy_by = (by_type[1] == NODE) ? sy_node : sy_cell;
    const double* sy_bz = (bz_type[1] == NODE) ? sy_node_v : sy_cell_v;
    int const k_ex = (ex_type[1] == NODE) ? k_node : k_cell;
    int const k_ey = (ey_type[1] == NODE) ? k_node_v : k_cell_v;
    int const k_ez = (ez_type[1] == NODE) ? k_node : k_cell;
    int const k_bx = (bx_type[1] == NODE) ? k_node_v : k_cell_v;
    int const k_by = (by_type[1] == NODE) ? k_node : k_cell;
    int const k_bz = (bz_type[1] == NODE) ? k_node_v : k_cell_v;

    // z direction
    const double z = (zp - xyzmin[2]) * dinv[2];
    double sz_node[depos_order + 1];
    double sz_cell[depos_order + 1];
    double sz_node_v[NG];
    double sz_cell_v[NG];
    int l_node = 0, l_cell = 0, l_node_v = 0, l_cell_v = 0;
    if ((ex_type[zdir] == NODE) || (ey_type[zdir] == NODE) || (bz_type[zdir] == NODE)) {
        l_node = compute_shape_factor<depos_order>(sz_node, z);
    }
    if ((ex_type[zdir] == CELL) || (ey_type[zdir] == CELL) || (bz_type[zdir] == CELL)) {
        l_cell = compute_shape_factor<depos_order>(sz_cell, z - 0.5);
    }
    if ((ez_type[zdir] == NODE) || (bx_type[zdir] == NODE) || (by_type[zdir] == NODE)) {
        l_node_v = compute_shape_factor<depos_order - galerkin_interpolation>(sz_node_v, z);
    }
    if ((ez_type[zdir] == CELL) || (bx_type[zdir] == CELL) || (by_type[zdir] == CELL)) {
        l_cell_v = compute_shape_factor<depos_order - galerkin_interpolation>(sz_cell_v, z - 0.5);
    }
    const double* sz_ex = (ex_type[zdir] == NODE) ? sz_node : sz_cell;
    const double* sz_ey = (ey_type[zdir] == NODE) ? sz_node : sz_cell;
    const double* sz_ez = (ez_type[zdir] == NODE) ? sz_node_v : sz_cell_v;
    const double* sz_bx = (bx_type[zdir] == NODE) ? sz_node_v : sz_cell_v;
    const double* sz_by = (by_type[zdir] == NODE) ? sz_node_v : sz_cell_v;
    const double* sz_bz = (bz_type[zdir] == NODE) ? sz_node : sz_cell;
    int const l_ex = (ex_type[zdir] == NODE) ? l_node : l_cell;
    int const l_ey = (ey_type[zdir] == NODE) ? l_node : l_cell;
    int const l_ez = (ez_type[zdir] == NODE) ? l_node_v : l_cell_v;
    int const l_bx = (bx_type[zdir] == NODE) ? l_node_v : l_cell_v;
    int const l_by = (by_type[zdir] == NODE) ? l_node_v : l_cell_v;
    int const l_bz = (bz_type[zdir] == NODE) ? l_node : l_cell;

    constexpr int O = depos_order;
    constexpr int G = depos_order - galerkin_interpolation;
    // FieldGather.H:368-423, component order Ex,Ey,Ez,Bz,By,Bx
    for (int iz = 0; iz <= O; iz++)
        for (int iy = 0; iy <= O; iy++)
            for (int ix = 0; ix <= G; ix++)
                Exp += sx_ex[ix] * sy_ex[iy] * sz_ex[iz] *
                       ex_arr(lo[0] + j_ex + ix, lo[1] + k_ex + iy, lo[2] + l_ex + iz);
    for (int iz = 0; iz <= O; iz++)
        for (int iy = 0; iy <= G; iy++)
            for (int ix = 0; ix <= O; ix++)
                Eyp += sx_ey[ix] * sy_ey[iy] * sz_ey[iz] *
                       ey_arr(lo[0] + j_ey + ix, lo[1] + k_ey + iy, lo[2] + l_ey + iz);
    for (int iz = 0; iz <= G; iz++)
        for (int iy = 0; iy <= O; iy++)
            for (int ix = 0; ix <= O; ix++)
                Ezp += sx_ez[ix] * sy_ez[iy] * sz_ez[iz] *
                       ez_arr(lo[0] + j_ez + ix, lo[1] + k_ez + iy, lo[2] + l_ez + iz);
    for (int iz = 0; iz <= O; iz++)
        for (int iy = 0; iy <= G; iy++)
            for (int ix = 0; ix <= G; ix++)
                Bzp += sx_bz[ix] * sy_bz[iy] * sz_bz[iz] *
                       bz_arr(lo[0] + j_bz + ix, lo[1] + k_bz + iy, lo[2] + l_bz + iz);
    for (int iz = 0; iz <= G; iz++)
        for (int iy = 0; iy <= O; iy++)
            for (int ix = 0; ix <= G; ix++)
                Byp += sx_by[ix] * sy_by[iy] * sz_by[iz] *
                       by_arr(lo[0] + j_by + ix, lo[1] + k_by + iy, lo[2] + l_by + iz);
    for (int iz = 0; iz <= G; iz++)
        for (int iy = 0; iy <= G; iy++)
            for (int ix = 0; ix <= O; ix++)
                Bxp += sx_bx[ix] * sy_bx[iy] * sz_bx[iz] *
                       bx_arr(lo[0] + j_bx + ix, lo[1] + k_bx + iy, lo[2] + l_bx + iz);
}

// ---------------------------------------------------------------------------
// Source/Particles/Pusher/UpdateMomentumBoris.H:15-53
inline void UpdateMomentumBoris(double& ux, double& uy, double& uz,
                                const double Ex, const double Ey, const double Ez,
                                const double Bx, const double By, const double Bz,
                                const double q, const double m, const double dt) {
    const double econst = 0.5 * q * dt / m;
    ux += econst * Ex;
    uy += econst * Ey;
    uz += econst * Ez;
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = 1. / std::sqrt(1. + (ux * ux + uy * uy + uz * uz) * inv_c2);
    const double tx = econst * inv_gamma * Bx;
    const double ty = econst * inv_gamma * By;
    const double tz = econst * inv_gamma * Bz;
    const double tsqi = 2. / (1. + tx * tx + ty * ty + tz * tz);
    const double sx = tx * tsqi;
    const double sy = ty * tsqi;
    const double sz = tz * tsqi;
    const double ux_p = ux + uy * tz - uz * ty;
    const double uy_p = uy + uz * tx - ux * tz;
    const double uz_p = uz + ux * ty - uy * tx;
    ux += uy_p * sz - uz_p * sy;
    uy += uz_p * sx - ux_p * sz;
    uz += ux_p * sy - uy_p * sx;
    ux += econst * Ex;
    uy += econst * Ey;
    uz += econst * Ez;
}

// Source/Particles/Pusher/UpdateMomentumVay.H:19-62
inline void UpdateMomentumVay(double& ux, double& uy, double& uz,
                              const double Ex, const double Ey, const double Ez,
                              const double Bx, const double By, const double Bz,
                              const double q, const double m, const double dt) {
    const double econst = q * dt / m;
    const double bconst = 0.5 * q * dt / m;
    constexpr double invclight = 1. / PhysConst::c;
    constexpr double invclightsq = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = 1. / std::sqrt(1. + (ux * ux + uy * uy + uz * uz) * invclightsq);
    const double taux = bconst * Bx;
    const double tauy = bconst * By;
    const double tauz = bconst * Bz;
    const double tausq = taux * taux + tauy * tauy + tauz * tauz;
    const double uxpr = ux + econst * Ex + (uy * tauz - uz * tauy) * inv_gamma;
    const double uypr = uy + econst * Ey + (uz * taux - ux * tauz) * inv_gamma;
    const double uzpr = uz + econst * Ez + (ux * tauy - uy * taux) * inv_gamma;
    const double gprsq = (1. + (uxpr * uxpr + uypr * uypr + uzpr * uzpr) * invclightsq);
    const double ust = (uxpr * taux + uypr * tauy + uzpr * tauz) * invclight;
    const double sigma = gprsq - tausq;
    const double gisq = 2. / (sigma + std::sqrt(sigma * sigma + 4. * (tausq + ust * ust)));
    const double bg = bconst * std::sqrt(gisq);
    const double tx = bg * Bx;
    const double ty = bg * By;
    const double tz = bg * Bz;
    const double s = 1. / (1. + tausq * gisq);
    const double tu = tx * uxpr + ty * uypr + tz * uzpr;
    ux = s * (uxpr + tx * tu + uypr * tz - uzpr * ty);
    uy = s * (uypr + ty * tu + uzpr * tx - uxpr * tz);
    uz = s * (uzpr + tz * tu + uxpr * ty - uypr * tx);
}

// Source/Particles/Pusher/UpdateMomentumHigueraCary.H:20-68
inline void UpdateMomentumHigueraCary(double& ux, double& uy, double& uz,
                                      const double Ex, const double Ey, const double Ez,
                                      const double Bx, const double By, const double Bz,
                                      const double q, const double m, const double dt) {
    const double qmt = 0.5 * q * dt / m;
    constexpr double invclight = 1. / PhysConst::c;
    constexpr double invclightsq = 1. / (PhysConst::c * PhysConst::c);
    const double umx = ux + qmt * Ex;
    const double umy = uy + qmt * Ey;
    const double umz = uz + qmt * Ez;
    double gamma = 1. + (umx * umx + umy * umy + umz * umz) * invclightsq;
    const double betax = qmt * Bx;
    const double betay = qmt * By;
    const double betaz = qmt * Bz;
    const double betam = betax * betax + betay * betay + betaz * betaz;
    const double sigma = gamma - betam;
    const double ust = (umx * betax + umy * betay + umz * betaz) * invclight;
    gamma = 1. / std::sqrt(0.5 * (sigma + std::sqrt(sigma * sigma + 4. * (betam + ust * ust))));
    const double tx = gamma * betax;
    const double ty = gamma * betay;
    const double tz = gamma * betaz;
    const double s = 1. / (1. + (tx * tx + ty * ty + tz * tz));
    const double umt = umx * tx + umy * ty + umz * tz;
    const double upx = s * (umx + umt * tx + umy * tz - umz * ty);
    const double upy = s * (umy + umt * ty + umz * tx - umx * tz);
    const double upz = s * (umz + umt * tz + umx * ty - umy * tx);
    ux = upx + qmt * Ex + upy * tz - upz * ty;
    uy = upy + qmt * Ey + upz * tx - upx * tz;
    uz = upz + qmt * Ez + upx * ty - upy * tx;
}

// Source/Particles/Pusher/UpdateMomentumBorisWithRadiationReaction.H:20-93
inline void UpdateMomentumBorisWithRadiationReaction(double& ux, double& uy, double& uz,
                                                     const double Ex, const double Ey, const double Ez,
                                                     const double Bx, const double By, const double Bz,
                                                     const double q, const double m, const double dt) {
    const double ux_old = ux, uy_old = uy, uz_old = uz;
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    UpdateMomentumBoris(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
    const double ux_n = (ux + ux_old) * 0.5;
    const double uy_n = (uy + uy_old) * 0.5;
    const double uz_n = (uz + uz_old) * 0.5;
    const double gamma_n = std::sqrt(1. + (ux_n * ux_n + uy_n * uy_n + uz_n * uz_n) * inv_c2);
    const double inv_gamma_n = 1.0 / gamma_n;
    const double vx_n = ux_n * inv_gamma_n;
    const double vy_n = uy_n * inv_gamma_n;
    const double vz_n = uz_n * inv_gamma_n;
    const double bx_n = vx_n / PhysConst::c;
    const double by_n = vy_n / PhysConst::c;
    const double bz_n = vz_n / PhysConst::c;
    const double flx_q = (Ex + vy_n * Bz - vz_n * By);
    const double fly_q = (Ey + vz_n * Bx - vx_n * Bz);
    const double flz_q = (Ez + vx_n * By - vy_n * Bx);
    const double fl_q2 = flx_q * flx_q + fly_q * fly_q + flz_q * flz_q;
    const double bdotE = (bx_n * Ex + by_n * Ey + bz_n * Ez);
    const double bdotE2 = bdotE * bdotE;
    const double coeff = gamma_n * gamma_n * (fl_q2 - bdotE2);
    const double q_over_mc = q / (m * PhysConst::c);
    const double RRcoeff = (2.0 / 3.0) * PhysConst::r_e * q_over_mc * q_over_mc;
    const double frx = RRcoeff * (PhysConst::c * (fly_q * Bz - flz_q * By) + bdotE * Ex - coeff * bx_n);
    const double fry = RRcoeff * (PhysConst::c * (flz_q * Bx - flx_q * Bz) + bdotE * Ey - coeff * by_n);
    const double frz = RRcoeff * (PhysConst::c * (flx_q * By - fly_q * Bx) + bdotE * Ez - coeff * bz_n);
    ux += frx * dt;
    uy += fry * dt;
    uz += frz * dt;
}

// Source/Particles/Pusher/PushSelector.H:38-102 (Boris / Vay / Higuera-Cary / radiation-reaction branches; ion_lev = 1)
inline void doParticleMomentumPush(double& ux, double& uy, double& uz,
                                   const double Ex, const double Ey, const double Ez,
                                   const double Bx, const double By, const double Bz,
                                   const double m, const double a_q, const int pusher_algo,
                                   const double dt) {
    double qp = a_q;
    qp *= 1;  // ion_lev ? ion_lev[ip] : 1
    if (pusher_algo == WXA_PUSHER_BORIS) {
        UpdateMomentumBoris(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, qp, m, dt);
    } else if (pusher_algo == WXA_PUSHER_VAY) {
        UpdateMomentumVay(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, qp, m, dt);
    } else if (pusher_algo == WXA_PUSHER_HC) {
        UpdateMomentumHigueraCary(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, qp, m, dt);
    } else if (pusher_algo == WXA_PUSHER_BORIS_RR) {   // do_crr
        UpdateMomentumBorisWithRadiationReaction(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, qp, m, dt);
    }
}

// Source/Particles/Pusher/UpdatePosition.H:24-45
inline void UpdatePosition(double& x, double& y, double& z,
                           const double ux, const double uy, const double uz, const double dt) {
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = 1. / std::sqrt(1. + (ux * ux + uy * uy + uz * uz) * inv_c2);
    x += ux * inv_gamma * dt;
    y += uy * inv_gamma * dt;
    z += uz * inv_gamma * dt;
}

// ---------------------------------------------------------------------------
// Source/Particles/Deposition/CurrentDeposition.H:48-249 (3-D branch), one particle.
template <int depos_order>
inline void doDepositionShapeNKernel(const double xp, const double yp, const double zp,
                                     const double wq, const double vx, const double vy, const double vz,
                                     const Arr& jx_arr, const Arr& jy_arr, const Arr& jz_arr,
                                     const int* jx_type, const int* jy_type, const int* jz_type,
                                     const double relative_time, const double* dinv,
                                     const double* xyzmin, const double invvol, const int* lo) {
    const double wqx = wq * invvol * vx;
    const double wqy = wq * invvol * vy;
    const double wqz = wq * invvol * vz;

    const double xmid = ((xp - xyzmin[0]) + relative_time * vx) * dinv[0];
    double sx_node[depos_order + 1] = {0.};
    double sx_cell[depos_order + 1] = {0.};
    int j_node = 0, j_cell = 0;
    if (jx_type[0] == NODE || jy_type[0] == NODE || jz_type[0] == NODE) {
        j_node = compute_shape_factor<depos_order>(sx_node, xmid);
    }
    if (jx_type[0] == CELL || jy_type[0] == CELL || jz_type[0] == CELL) {
        j_cell = compute_shape_factor<depos_order>(sx_cell, xmid - 0.5);
    }
    double sx_jx[depos_order + 1], sx_jy[depos_order + 1], sx_jz[depos_order + 1];
    for (int ix = 0; ix <= depos_order; ix++) {
        sx_jx[ix] = (jx_type[0] == NODE) ? sx_node[ix] : sx_cell[ix];
        sx_jy[ix] = (jy_type[0] == NODE) ? sx_node[ix] : sx_cell[ix];
        sx_jz[ix] = (jz_type[0] == NODE) ? sx_node[ix] : sx_cell[ix];
    }
    int const j_jx = (jx_type[0] == NODE) ? j_node : j_cell;
    int const j_jy = (jy_type[0] == NODE) ? j_node : j_cell;
    int const j_jz = (jz_type[0] == NODE) ? j_node : j_cell;

    const double ymid = ((yp - xyzmin[1]) + relative_time * vy) * dinv[1];
    double sy_node[depos_order + 1] = {0.};
    double sy_cell[depos_order + 1] = {0.};
    int k_node = 0, k_cell = 0;
    if (jx_type[1] == NODE || jy_type[1] == NODE || jz_type[1] == NODE) {
        k_node = compute_shape_factor<depos_order>(sy_node, ymid);
    }
    if (jx_type[1] == CELL || jy_type[1] == CELL || jz_type[1] == CELL) {
        k_cell = compute_shape_factor<depos_order>(sy_cell, ymid - 0.5);
    }
    double sy_jx[depos_order + 1], sy_jy[depos_order + 1], sy_jz[depos_order + 1];
    for (int iy = 0; iy <= depos_order; iy++) {
        sy_jx[iy] = (jx_type[1] == NODE) ? sy_node[iy] : sy_cell[iy];
        sy_jy[iy] = (jy_type[1] == NODE) ? sy_node[iy] : sy_cell[iy];
        sy_jz[iy] = (jz_type[1] == NODE) ? sy_node[iy] : sy_cell[iy];
    }
    int const k_jx = (jx_type[1] == NODE) ? k_node : k_cell;
    int const k_jy = (jy_type[1] == NODE) ? k_node : k_cell;
    int const k_jz = (jz_type[1] == NODE) ? k_node : k_cell;

    const double zmid = ((zp - xyzmin[2]) + relative_time * vz) * dinv[2];
    double sz_node[depos_order + 1] = {0.};
    double sz_cell[depos_order + 1] = {0.};
    int l_node = 0, l_cell = 0;
    if (jx_type[2] == NODE || jy_type[2] == NODE || jz_type[2] == NODE) {
        l_node = compute_shape_factor<depos_order>(sz_node, zmid);
    }
    if (jx_type[2] == CELL || jy_type[2] == CELL || jz_type[2] == CELL) {
        l_cell = compute_shape_factor<depos_order>(sz_cell, zmid - 0.5);
    }
    double sz_jx[depos_order + 1], sz_jy[depos_order + 1], sz_jz[depos_order + 1];
    for (int iz = 0; iz <= depos_order; iz++) {
        sz_jx[iz] = (jx_type[2] == NODE) ? sz_node[iz] : sz_cell[iz];
        sz_jy[iz] = (jy_type[2] == NODE) ? sz_node[iz] : sz_cell[iz];
        sz_jz[iz] = (jz_type[2] == NODE) ? sz_node[iz] : sz_cell[iz];
    }
    int const l_jx = (jx_type[2] == NODE) ? l_node : l_cell;
    int const l_jy = (jy_type[2] == NODE) ? l_node : l_cell;
    int const l_jz = (jz_type[2] == NODE) ? l_node : l_cell;

    for (int iz = 0; iz <= depos_order; iz++) {
        for (int iy = 0; iy <= depos_order; iy++) {
            for (int ix = 0; ix <= depos_order; ix++) {
                jx_arr(lo[0] + j_jx + ix, lo[1] + k_jx + iy, lo[2] + l_jx + iz) +=
                    sx_jx[ix] * sy_jx[iy] * sz_jx[iz] * wqx;
                jy_arr(lo[0] + j_jy + ix, lo[1] + k_jy + iy, lo[2] + l_jy + iz) +=
                    sx_jy[ix] * sy_jy[iy] * sz_jy[iz] * wqy;
                jz_arr(lo[0] + j_jz + ix, lo[1] + k_jz + iy, lo[2] + l_jz + iz) +=
                    sx_jz[ix] * sy_jz[iy] * sz_jz[iz] * wqz;
            }
        }
    }
}

// Source/Particles/Deposition/CurrentDeposition.H:273-335, per-particle body (:309-334)
template <int depos_order>
inline void doDepositionShapeN_one(const double xp, const double yp, const double zp, const double w,
                                   const double ux, const double uy, const double uz,
                                   const Arr& jx, const Arr& jy, const Arr& jz,
                                   const int* jx_type, const int* jy_type, const int* jz_type,
                                   double relative_time, const double* dinv, const double* xyzmin,
                                   const int* lo, double q) {
    const double invvol = dinv[0] * dinv[1] * dinv[2];
    const double clightsq = 1.0 / PhysConst::c / PhysConst::c;
    const double gaminv = 1.0 / std::sqrt(1.0 + ux * ux * clightsq + uy * uy * clightsq + uz * uz * clightsq);
    const double vx = ux * gaminv;
    const double vy = uy * gaminv;
    const double vz = uz * gaminv;
    const double wq = q * w;
    doDepositionShapeNKernel<depos_order>(xp, yp, zp, wq, vx, vy, vz, jx, jy, jz, jx_type, jy_type,
                                          jz_type, relative_time, dinv, xyzmin, invvol, lo);
}

// Source/Particles/Deposition/CurrentDeposition.H:642-907, per-particle body, 3-D branch (:683-824)
template <int depos_order>
inline void doEsirkepovDepositionShapeN_one(const double xp, const double yp, const double zp,
                                            const double w, const double uxp, const double uyp,
                                            const double uzp, const Arr& Jx_arr, const Arr& Jy_arr,
                                            const Arr& Jz_arr, double dt, double relative_time,
                                            const double* dinv, const double* xyzmin, const int* lo,
                                            double q) {
    const double invdtd_x = (1.0 / dt) * dinv[1] * dinv[2];
    const double invdtd_y = (1.0 / dt) * dinv[0] * dinv[2];
    const double invdtd_z = (1.0 / dt) * dinv[0] * dinv[1];
    constexpr double clightsq = 1.0 / (PhysConst::c * PhysConst::c);
    constexpr double one_third = 1.0 / 3.0;
    constexpr double one_sixth = 1.0 / 6.0;

    double const gaminv = 1.0 / std::sqrt(1.0 + uxp * uxp * clightsq + uyp * uyp * clightsq + uzp * uzp * clightsq);
    double const wq = q * w;

    double const x_new = (xp - xyzmin[0] + (relative_time + 0.5 * dt) * uxp * gaminv) * dinv[0];
    double const x_old = x_new - dt * dinv[0] * uxp * gaminv;
    double const y_new = (yp - xyzmin[1] + (relative_time + 0.5 * dt) * uyp * gaminv) * dinv[1];
    double const y_old = y_new - dt * dinv[1] * uyp * gaminv;
    double const z_new = (zp - xyzmin[2] + (relative_time + 0.5 * dt) * uzp * gaminv) * dinv[2];
    double const z_old = z_new - dt * dinv[2] * uzp * gaminv;

    double sx_new[depos_order + 3] = {0.};
    double sx_old[depos_order + 3] = {0.};
    const int i_new = compute_shape_factor<depos_order>(sx_new + 1, x_new);
    const int i_old = compute_shifted_shape_factor<depos_order>(sx_old, x_old, i_new);
    double sy_new[depos_order + 3] = {0.};
    double sy_old[depos_order + 3] = {0.};
    const int j_new = compute_shape_factor<depos_order>(sy_new + 1, y_new);
    const int j_old = compute_shifted_shape_factor<depos_order>(sy_old, y_old, j_new);
    double sz_new[depos_order + 3] = {0.};
    double sz_old[depos_order + 3] = {0.};
    const int k_new = compute_shape_factor<depos_order>(sz_new + 1, z_new);
    const int k_old = compute_shifted_shape_factor<depos_order>(sz_old, z_old, k_new);

    int dil = 1, diu = 1;
    if (i_old < i_new) { dil = 0; }
    if (i_old > i_new) { diu = 0; }
    int djl = 1, dju = 1;
    if (j_old < j_new) { djl = 0; }
    if (j_old > j_new) { dju = 0; }
    int dkl = 1, dku = 1;
    if (k_old < k_new) { dkl = 0; }
    if (k_old > k_new) { dku = 0; }

    for (int k = dkl; k <= depos_order + 2 - dku; k++) {
        for (int j = djl; j <= depos_order + 2 - dju; j++) {
            double sdxi = 0.;
            for (int i = dil; i <= depos_order + 1 - diu; i++) {
                sdxi += wq * invdtd_x * (sx_old[i] - sx_new[i]) *
                        (one_third * (sy_new[j] * sz_new[k] + sy_old[j] * sz_old[k]) +
                         one_sixth * (sy_new[j] * sz_old[k] + sy_old[j] * sz_new[k]));
                Jx_arr(lo[0] + i_new - 1 + i, lo[1] + j_new - 1 + j, lo[2] + k_new - 1 + k) += sdxi;
            }
        }
    }
    for (int k = dkl; k <= depos_order + 2 - dku; k++) {
        for (int i = dil; i <= depos_order + 2 - diu; i++) {
            double sdyj = 0.;
            for (int j = djl; j <= depos_order + 1 - dju; j++) {
                sdyj += wq * invdtd_y * (sy_old[j] - sy_new[j]) *
                        (one_third * (sx_new[i] * sz_new[k] + sx_old[i] * sz_old[k]) +
                         one_sixth * (sx_new[i] * sz_old[k] + sx_old[i] * sz_new[k]));
                Jy_arr(lo[0] + i_new - 1 + i, lo[1] + j_new - 1 + j, lo[2] + k_new - 1 + k) += sdyj;
            }
        }
    }
    for (int j = djl; j <= depos_order + 2 - dju; j++) {
        for (int i = dil; i <= depos_order + 2 - diu; i++) {
            double sdzk = 0.;
            for (int k = dkl; k <= depos_order + 1 - dku; k++) {
                sdzk += wq * invdtd_z * (sz_old[k] - sz_new[k]) *
                        (one_third * (sx_new[i] * sy_new[j] + sx_old[i] * sy_old[j]) +
                         one_sixth * (sx_new[i] * sy_old[j] + sx_old[i] * sy_new[j]));
                Jz_arr(lo[0] + i_new - 1 + i, lo[1] + j_new - 1 + j, lo[2] + k_new - 1 + k) += sdzk;
            }
        }
    }
}

// Source/Particles/Deposition/ChargeDeposition.H:37-180, per-particle body, 3-D branch
template <int depos_order>
inline void doChargeDepositionShapeN_one(const double xp, const double yp, const double zp,
                                         const double w, const Arr& rho_arr, const int* rho_type,
                                         const double* dinv, const double* xyzmin, const int* lo,
                                         double q) {
    const double invvol = dinv[0] * dinv[1] * dinv[2];
    const double wq = q * w * invvol;
    const double x = (xp - xyzmin[0]) * dinv[0];
    double sx[depos_order + 1] = {0.};
    int i = 0;
    if (rho_type[0] == NODE) { i = compute_shape_factor<depos_order>(sx, x); }
    else { i = compute_shape_factor<depos_order>(sx, x - 0.5); }
    const double y = (yp - xyzmin[1]) * dinv[1];
    double sy[depos_order + 1] = {0.};
    int j = 0;
    if (rho_type[1] == NODE) { j = compute_shape_factor<depos_order>(sy, y); }
    else { j = compute_shape_factor<depos_order>(sy, y - 0.5); }
    const double z = (zp - xyzmin[2]) * dinv[2];
    double sz[depos_order + 1] = {0.};
    int k = 0;
    if (rho_type[2] == NODE) { k = compute_shape_factor<depos_order>(sz, z); }
    else { k = compute_shape_factor<depos_order>(sz, z - 0.5); }
    for (int iz = 0; iz <= depos_order; iz++)
        for (int iy = 0; iy <= depos_order; iy++)
            for (int ix = 0; ix <= depos_order; ix++)
                rho_arr(lo[0] + i + ix, lo[1] + j + iy, lo[2] + k + iz) += sx[ix] * sy[iy] * sz[iz] * wq;
}

}  // namespace orc
#endif
