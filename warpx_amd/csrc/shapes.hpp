// Device restatement of the particle arithmetic (shape factors, gather on the Yee
// grid, Boris/Vay pushers, position update).  Formulas follow the reference headers
// cited at each function; the code organisation (Yee-specialised gather with two shape
// arrays per direction, register-resident weights) is this project's own.
#ifndef WXA_SHAPES_HPP_
#define WXA_SHAPES_HPP_

#include "common.hpp"

namespace wxa {

// B-spline weights of the offset xi from the reference node, Source/Particles/ShapeFactors.H:27-84.
// RN = false: the device compiler may contract a*b+c into FMAs (gather: one more correct bit, fewer instructions).
// RN = true: one rounding per operation, the CPU path's arithmetic.  The Esirkepov deposition needs it: it
// subtracts the weights of the old and the new position, and the contraction choices of the compiler differ between
// the two evaluations (they depend on how each weight is consumed), so that identical positions would give weights
// that differ by an ulp -- a spurious current where the reference deposits exactly 0.
#define WXA_BSPLINE_BODY                                                   \
    if constexpr (ORDER == 0) {                                            \
        s[0] = 1.0;                                                        \
    } else if constexpr (ORDER == 1) {                                     \
        s[0] = 1.0 - xi;                                                   \
        s[1] = xi;                                                         \
    } else if constexpr (ORDER == 2) {                                     \
        s[0] = 0.5 * (0.5 - xi) * (0.5 - xi);                              \
        s[1] = 0.75 - xi * xi;                                             \
        s[2] = 0.5 * (0.5 + xi) * (0.5 + xi);                              \
    } else if constexpr (ORDER == 3) {                                     \
        const double om = 1.0 - xi;                                        \
        s[0] = (1.0 / 6.0) * om * om * om;                                 \
        s[1] = (2.0 / 3.0) - xi * xi * (1.0 - xi / 2.0);                   \
        s[2] = (2.0 / 3.0) - om * om * (1.0 - 0.5 * om);                   \
        s[3] = (1.0 / 6.0) * xi * xi * xi;                                 \
    } else {   /* quartic spline, ShapeFactors.H:67-77 */                  \
        static_assert(ORDER == 4, "orders 0..4");                          \
        const double lo = 0.5 - xi, hi = 0.5 + xi;                         \
        s[0] = (1.0 / 24.0) * lo * lo * lo * lo;                           \
        s[1] = (1.0 / 24.0) * (4.75 - 11.0 * xi + 4.0 * xi * xi * (1.5 + xi - xi * xi)); \
        s[2] = (1.0 / 24.0) * (14.375 + 6.0 * xi * xi * (xi * xi - 2.5));  \
        s[3] = (1.0 / 24.0) * (4.75 + 11.0 * xi + 4.0 * xi * xi * (1.5 - xi - xi * xi)); \
        s[4] = (1.0 / 24.0) * hi * hi * hi * hi;                           \
    }

template <int ORDER, bool RN>
__device__ __forceinline__ void bspline_weights(double* __restrict__ s, const double x, const int j) {
    if constexpr (RN) {
#pragma clang fp contract(off)
        const double xi = x - (double)j;
        WXA_BSPLINE_BODY
    } else {
        const double xi = x - (double)j;
        WXA_BSPLINE_BODY
    }
}
#undef WXA_BSPLINE_BODY

// reference node of a coordinate: nearest node for even orders, node below for odd ones.
// x >= 0 is guaranteed by the guard-grown index origin, so truncation == floor.
template <int ORDER>
__device__ __forceinline__ int shape_node_of(const double x) {
    if constexpr (ORDER % 2 == 0) return (int)(x + 0.5);
    else return (int)x;
}

// Compute_shape_factor, ShapeFactors.H:27-84.  Returns the leftmost index.
template <int ORDER, bool RN = false>
__device__ __forceinline__ int shape_factor(double* __restrict__ s, const double x) {
    const int j = shape_node_of<ORDER>(x);
    bspline_weights<ORDER, RN>(s, x, j);
    return j - ORDER / 2;   // j, j - 1, j - 1, j - 2 for orders 1 .. 4
}

// The same weights with the reference node j imposed by the caller (xint = x - j) instead of
// derived from x: used for the old position of a particle that the caller has classified as
// "same cell as the new position".  When x sits on a cell boundary to within an ulp the two
// evaluations of (int)x can disagree; the spline pieces are continuous there, so imposing j
// changes the weights by O(ulp) while keeping them on the slots the caller expects.
template <int ORDER, bool RN = false>
__device__ __forceinline__ void shape_weights_at(double* __restrict__ s, const double x, const int j) {
    static_assert(ORDER >= 1 && ORDER <= 4, "orders 1..4");
    bspline_weights<ORDER, RN>(s, x, j);
}

// reference node j of shape_factor<ORDER> from its return value (leftmost index)
template <int ORDER>
__device__ __forceinline__ int shape_node(int leftmost) { return leftmost + ORDER / 2; }

// Old-position weights on the slots of the new position (Esirkepov),
// Source/Particles/ShapeFactors.H:93-156.  s has ORDER+3 entries, all written here.
// The reference stores the ORDER+1 weights at the run-time offset 1+i_shift; a run-time
// register index would spill the array to scratch memory on gfx950, so every slot is
// selected from the (at most three) candidate weights instead.
template <int ORDER, bool RN = false>
__device__ __forceinline__ int shifted_shape_factor(double* __restrict__ s, const double x_old,
                                                    const int i_new) {
    static_assert(ORDER >= 1 && ORDER <= 4, "orders 1..4");
    double w[ORDER + 1];
    // ORDER 1 floors (ShapeFactors.H:112), the others truncate like Compute_shape_factor
    const int i = ORDER == 1 ? (int)floor(x_old) : shape_node_of<ORDER>(x_old);
    const int sh = i - (i_new + ORDER / 2);   // :113, :122, :134, :147
    const int ret = i - ORDER / 2;
    bspline_weights<ORDER, RN>(w, x_old, i);
    // slot a holds w[a - 1 - sh] when that index exists (sh in {-1,0,+1} under the CFL limit)
#pragma unroll
    for (int a = 0; a < ORDER + 3; ++a) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b <= ORDER; ++b) v = (a - 1 - sh == b) ? w[b] : v;
        s[a] = v;
    }
    return ret;
}

// 1/sqrt(x) as one reciprocal-square-root evaluation (v_rsq_f64 + refinement, 12 instructions,
// within 1 ulp) instead of a correctly rounded square root followed by a correctly rounded
// division (34 instructions).  The Lorentz factor is taken once per particle in the pusher, the
// position update and the deposition: ~10 % of the gather's and ~5 % of the deposition's arithmetic.
__device__ __forceinline__ double inv_sqrt(const double x) { return rsqrt(x); }

// Source/Particles/Pusher/UpdateMomentumBoris.H:15-53
__device__ __forceinline__ void push_boris(double& ux, double& uy, double& uz, const double Ex,
                                           const double Ey, const double Ez, const double Bx,
                                           const double By, const double Bz, const double q,
                                           const double m, const double dt) {
    const double econst = 0.5 * q * dt / m;
    ux += econst * Ex; uy += econst * Ey; uz += econst * Ez;
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = inv_sqrt(1. + (ux * ux + uy * uy + uz * uz) * inv_c2);
    const double tx = econst * inv_gamma * Bx;
    const double ty = econst * inv_gamma * By;
    const double tz = econst * inv_gamma * Bz;
    const double tsqi = 2. / (1. + tx * tx + ty * ty + tz * tz);
    const double sx = tx * tsqi, sy = ty * tsqi, sz = tz * tsqi;
    const double ux_p = ux + uy * tz - uz * ty;
    const double uy_p = uy + uz * tx - ux * tz;
    const double uz_p = uz + ux * ty - uy * tx;
    ux += uy_p * sz - uz_p * sy;
    uy += uz_p * sx - ux_p * sz;
    uz += ux_p * sy - uy_p * sx;
    ux += econst * Ex; uy += econst * Ey; uz += econst * Ez;
}

// Source/Particles/Pusher/UpdateMomentumVay.H:19-62
__device__ __forceinline__ void push_vay(double& ux, double& uy, double& uz, const double Ex,
                                         const double Ey, const double Ez, const double Bx,
                                         const double By, const double Bz, const double q,
                                         const double m, const double dt) {
    const double econst = q * dt / m;
    const double bconst = 0.5 * q * dt / m;
    constexpr double invclight = 1. / PhysConst::c;
    constexpr double invclightsq = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = inv_sqrt(1. + (ux * ux + uy * uy + uz * uz) * invclightsq);
    const double taux = bconst * Bx, tauy = bconst * By, tauz = bconst * Bz;
    const double tausq = taux * taux + tauy * tauy + tauz * tauz;
    const double uxpr = ux + econst * Ex + (uy * tauz - uz * tauy) * inv_gamma;
    const double uypr = uy + econst * Ey + (uz * taux - ux * tauz) * inv_gamma;
    const double uzpr = uz + econst * Ez + (ux * tauy - uy * taux) * inv_gamma;
    const double gprsq = (1. + (uxpr * uxpr + uypr * uypr + uzpr * uzpr) * invclightsq);
    const double ust = (uxpr * taux + uypr * tauy + uzpr * tauz) * invclight;
    const double sigma = gprsq - tausq;
    const double gisq = 2. / (sigma + sqrt(sigma * sigma + 4. * (tausq + ust * ust)));
    const double bg = bconst * sqrt(gisq);
    const double tx = bg * Bx, ty = bg * By, tz = bg * Bz;
    const double s = 1. / (1. + tausq * gisq);
    const double tu = tx * uxpr + ty * uypr + tz * uzpr;
    ux = s * (uxpr + tx * tu + uypr * tz - uzpr * ty);
    uy = s * (uypr + ty * tu + uzpr * tx - uxpr * tz);
    uz = s * (uzpr + tz * tu + uxpr * ty - uypr * tx);
}

// Source/Particles/Pusher/UpdateMomentumHigueraCary.H:20-68
__device__ __forceinline__ void push_hc(double& ux, double& uy, double& uz, const double Ex, const double Ey,
                                        const double Ez, const double Bx, const double By, const double Bz,
                                        const double q, const double m, const double dt) {
    const double qmt = 0.5 * q * dt / m;
    constexpr double invclight = 1. / PhysConst::c;
    constexpr double invclightsq = 1. / (PhysConst::c * PhysConst::c);
    const double umx = ux + qmt * Ex, umy = uy + qmt * Ey, umz = uz + qmt * Ez;
    double gamma = 1. + (umx * umx + umy * umy + umz * umz) * invclightsq;
    const double betax = qmt * Bx, betay = qmt * By, betaz = qmt * Bz;
    const double betam = betax * betax + betay * betay + betaz * betaz;
    const double sigma = gamma - betam;
    const double ust = (umx * betax + umy * betay + umz * betaz) * invclight;
    gamma = 1. / sqrt(0.5 * (sigma + sqrt(sigma * sigma + 4. * (betam + ust * ust))));
    const double tx = gamma * betax, ty = gamma * betay, tz = gamma * betaz;
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
__device__ __forceinline__ void push_boris_rr(double& ux, double& uy, double& uz, const double Ex, const double Ey,
                                              const double Ez, const double Bx, const double By, const double Bz,
                                              const double q, const double m, const double dt) {
    const double ux_old = ux, uy_old = uy, uz_old = uz;
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    push_boris(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
    const double ux_n = (ux + ux_old) * 0.5;
    const double uy_n = (uy + uy_old) * 0.5;
    const double uz_n = (uz + uz_old) * 0.5;
    const double gamma_n = sqrt(1. + (ux_n * ux_n + uy_n * uy_n + uz_n * uz_n) * inv_c2);
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

// doParticleMomentumPush (Source/Particles/Pusher/PushSelector.H:38-102), ion_lev = 1; the selector is a
// template parameter of the kernels (uniform per launch)
template <int PUSHER>
__device__ __forceinline__ void push_momentum(double& ux, double& uy, double& uz, const double Ex, const double Ey,
                                              const double Ez, const double Bx, const double By, const double Bz,
                                              const double q, const double m, const double dt) {
    if constexpr (PUSHER == WXA_PUSHER_BORIS) push_boris(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
    else if constexpr (PUSHER == WXA_PUSHER_VAY) push_vay(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
    else if constexpr (PUSHER == WXA_PUSHER_HC) push_hc(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
    else push_boris_rr(ux, uy, uz, Ex, Ey, Ez, Bx, By, Bz, q, m, dt);
}

// particles.E_external_particle / B_external_particle (constant): members of the container in the reference
// (m_E_external_particle, PhysicalParticleContainer.cpp:2589-2596,2705-2710).  The reference starts the gather sums
// from these values; here they are added to the gathered sums (same value, last-bit differences in the rounding).
// particles.*_ext_particle_init_style = repeated_plasma_lens (include/warpx_amd.h, wxa_repeated_plasma_lens);
// tab = starts | lengths | strengths_E | strengths_B, n entries each, device memory
struct ExtLens {
    int n;
    double period, time, dt, gamma_boost, uz_boost;
    const double* tab;
};
// fields[c * stride + ip], c = Ex Ey Bx By: per-particle external fields evaluated by a kernel of their own before the push
// (lens_fields_kernel, particles.hip); null = none.  Kept out of the gather kernels: with the lens arithmetic inlined
// the tile kernel went from 93 to 255 VGPRs and the global-memory one from 102 to 256.
struct ExtPerParticle {
    const double* fields;
    long stride;
};
struct ExtEB {
    double ex, ey, ez, bx, by, bz;
    ExtPerParticle pp;
};
__device__ __forceinline__ void add_external_fields(const ExtEB& ext, long ip, double& Ex, double& Ey, double& Ez, double& Bx,
                                                    double& By, double& Bz) {
    Ex += ext.ex; Ey += ext.ey; Ez += ext.ez; Bx += ext.bx; By += ext.by; Bz += ext.bz;
    // the gathered sums are complete here: without the pins the compiler sinks their tails past the branch below and the
    // register allocation of the 252-point gather falls apart (93 -> 247 VGPRs in the tile kernel)
    WXA_OPAQUE_F64(Ex); WXA_OPAQUE_F64(Ey); WXA_OPAQUE_F64(Ez); WXA_OPAQUE_F64(Bx); WXA_OPAQUE_F64(By); WXA_OPAQUE_F64(Bz);
    if (ext.pp.fields) {   // uniform
        Ex += ext.pp.fields[ip];
        Ey += ext.pp.fields[ext.pp.stride + ip];
        Bx += ext.pp.fields[2 * ext.pp.stride + ip];
        By += ext.pp.fields[3 * ext.pp.stride + ip];
    }
}

// GetExternalEBField::operator() (Source/Particles/Gather/GetExternalFields.H:137-189): the lens fields seen by a
// particle at (x, y, z) with momentum u before the push, added to the gathered fields
__device__ __forceinline__ void add_lens_fields(const ExtLens& L, const double x, const double y, const double z,
                                                const double ux, const double uy, const double uz, double& fEx,
                                                double& fEy, double& fBx, double& fBy) {
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    const double gamma = sqrt(1. + (ux * ux + uy * uy + uz * uz) * inv_c2);
    const double vzp = uz / gamma;
    double zl = z, zr = z + vzp * L.dt;
    if (L.gamma_boost > 1.) {
        zl = L.gamma_boost * zl + L.uz_boost * L.time;
        zr = L.gamma_boost * zr + L.uz_boost * (L.time + L.dt);
    }
    double Ex = 0., Ey = 0., Bx = 0., By = 0.;
    if (zl > 0) {
        const int i_lens = (int)floor(zl / L.period);
        if (i_lens < L.n) {
            const double lens_start = L.tab[i_lens] + i_lens * L.period;
            const double lens_end = lens_start + L.tab[L.n + i_lens];
            const double zl_bounded = fmin(fmax(zl, lens_start), lens_end);
            const double zr_bounded = fmin(fmax(zr, lens_start), lens_end);
            const double frac = (zr - zl) == 0. ? 1. : (zr_bounded - zl_bounded) / (zr - zl);
            const double sE = L.tab[2 * L.n + i_lens], sB = L.tab[3 * L.n + i_lens];
            Ex = x * frac * sE;
            Ey = y * frac * sE;
            Bx = +y * frac * sB;
            By = -x * frac * sB;
        }
    }
    if (L.gamma_boost > 1.) {
        const double Ex_boost = L.gamma_boost * Ex - L.uz_boost * By;
        const double Ey_boost = L.gamma_boost * Ey + L.uz_boost * Bx;
        const double Bx_boost = L.gamma_boost * Bx + L.uz_boost * Ey * inv_c2;
        const double By_boost = L.gamma_boost * By - L.uz_boost * Ex * inv_c2;
        Ex = Ex_boost; Ey = Ey_boost; Bx = Bx_boost; By = By_boost;
    }
    fEx += Ex; fEy += Ey; fBx += Bx; fBy += By;
}

// Source/Particles/Pusher/UpdatePosition.H:24-45
__device__ __forceinline__ void update_position(double& x, double& y, double& z, const double ux,
                                                const double uy, const double uz, const double dt) {
    constexpr double inv_c2 = 1. / (PhysConst::c * PhysConst::c);
    const double inv_gamma = inv_sqrt(1. + (ux * ux + uy * uy + uz * uz) * inv_c2);
    x += ux * inv_gamma * dt;
    y += uy * inv_gamma * dt;
    z += uz * inv_gamma * dt;
}

struct Geom {
    double xmin, ymin, zmin;
    double dxi, dyi, dzi;
    int lo0, lo1, lo2;
};
inline Geom make_geom(const wxa_grid_geom& g) {
    return Geom{g.xyzmin[0], g.xyzmin[1], g.xyzmin[2], g.dinv[0], g.dinv[1], g.dinv[2], g.lo[0], g.lo[1], g.lo[2]};
}

}  // namespace wxa
#endif
