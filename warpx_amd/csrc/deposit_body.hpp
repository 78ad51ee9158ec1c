// Per-particle current-deposition bodies, generic over the accumulation sink
// (global fp64 atomics, or an LDS tile).  Arithmetic follows
// Source/Particles/Deposition/CurrentDeposition.H:683-824 (Esirkepov, 3-D) and
// :48-249,309-334 (direct, 3-D) on the Yee grid.
#ifndef WXA_DEPOSIT_BODY_HPP_
#define WXA_DEPOSIT_BODY_HPP_

#include "shapes.hpp"

#include <type_traits>

namespace wxa {

struct ParticleState {
    double x, y, z, w, ux, uy, uz;
};

// Shape data of one particle for Esirkepov: weights at the new and the old position on
// the O+3 slots starting at grid index (lo + i_new - 1), plus the trimmed loop bounds.
template <int O>
struct EsirkepovShapes {
    double sx_new[O + 3], sx_old[O + 3], sy_new[O + 3], sy_old[O + 3], sz_new[O + 3], sz_old[O + 3];
    int bi, bj, bk;              // grid index of slot 0
    int dil, diu, djl, dju, dkl, dku;
    double wq;
};

// Grid coordinates of the particle at the start and at the end of the step, CurrentDeposition.H:700-716.
// The reference rounds x_new once and derives x_old from that rounded value, so a particle with u_d == 0 has
// x_old == x_new bit for bit, its old and new weights are identical and J_d is exactly 0 (the laser antenna's
// jx in test_3d_laser_injection.json).  With FMA contraction the device compiler would fuse the final "* dxi"
// into each consumer separately (fma(sum, dxi, -j) in the weights, a rounded product for the cell index and as
// the operand of x_old): old and new weights then differ by an ulp and deposit a spurious J of ~2e-16 of the
// transverse current.  This block keeps one rounding per operation, like the CPU path.
struct EsirkepovCoords {
    double x_new, x_old, y_new, y_old, z_new, z_old;
};
// What the Esirkepov arithmetic needs from (dt, relative_time, cell size), uniform per launch.  Evaluated once on the
// host, with the reference's operation order, and handed to the kernels as arguments: a uniform double computed in a
// kernel lives in a VGPR pair of every lane (there is no scalar fp64 unit), and the compiler hoists such values out
// of the particle loop -- a dozen of them cost the LDS-tile kernel 24 VGPRs through its whole life.
struct EsirkepovStep {
    double t_half;      // relative_time + 0.5 dt                      (CurrentDeposition.H:700)
    double dtdx[3];     // dt * dinv[d]                                 (:703)
    double invdtd[3];   // 1/dt * dinv[t1] * dinv[t2] per component     (:669-671)
};
inline EsirkepovStep make_esirkepov_step(const Geom& g, double dt, double relative_time) {
    EsirkepovStep es;
    es.t_half = relative_time + 0.5 * dt;
    es.dtdx[0] = dt * g.dxi; es.dtdx[1] = dt * g.dyi; es.dtdx[2] = dt * g.dzi;
    es.invdtd[0] = (1.0 / dt) * g.dyi * g.dzi;
    es.invdtd[1] = (1.0 / dt) * g.dxi * g.dzi;
    es.invdtd[2] = (1.0 / dt) * g.dxi * g.dyi;
    return es;
}
__device__ __forceinline__ EsirkepovCoords esirkepov_coords(const ParticleState& p, const Geom& g,
                                                            const EsirkepovStep& es) {
#pragma clang fp contract(off)
    constexpr double clightsq = 1.0 / (PhysConst::c * PhysConst::c);
    const double gaminv =
        inv_sqrt(1.0 + p.ux * p.ux * clightsq + p.uy * p.uy * clightsq + p.uz * p.uz * clightsq);
    EsirkepovCoords c;
    c.x_new = (p.x - g.xmin + es.t_half * p.ux * gaminv) * g.dxi;
    c.x_old = c.x_new - es.dtdx[0] * p.ux * gaminv;
    c.y_new = (p.y - g.ymin + es.t_half * p.uy * gaminv) * g.dyi;
    c.y_old = c.y_new - es.dtdx[1] * p.uy * gaminv;
    c.z_new = (p.z - g.zmin + es.t_half * p.uz * gaminv) * g.dzi;
    c.z_old = c.z_new - es.dtdx[2] * p.uz * gaminv;
    return c;
}

// old - new weight without contraction: the weights are products, and a fused fma(a, b, -s_new) would subtract the
// rounded new weight from the unrounded old one -- a residue of one rounding error where the reference has exactly 0
__device__ __forceinline__ double sub_rn(const double a, const double b) {
#pragma clang fp contract(off)
    return a - b;
}

template <int O>
__device__ __forceinline__ void esirkepov_shapes(const ParticleState& p, const Geom& g, double q,
                                                 const EsirkepovStep& es, EsirkepovShapes<O>& s) {
    s.wq = q * p.w;
    const EsirkepovCoords cc = esirkepov_coords(p, g, es);
    const double x_new = cc.x_new, x_old = cc.x_old, y_new = cc.y_new, y_old = cc.y_old, z_new = cc.z_new,
                 z_old = cc.z_old;
#pragma unroll
    for (int a = 0; a < O + 3; ++a) {
        s.sx_new[a] = 0.; s.sx_old[a] = 0.; s.sy_new[a] = 0.; s.sy_old[a] = 0.; s.sz_new[a] = 0.; s.sz_old[a] = 0.;
    }
    const int i_new = shape_factor<O, true>(s.sx_new + 1, x_new);
    const int i_old = shifted_shape_factor<O, true>(s.sx_old, x_old, i_new);
    const int j_new = shape_factor<O, true>(s.sy_new + 1, y_new);
    const int j_old = shifted_shape_factor<O, true>(s.sy_old, y_old, j_new);
    const int k_new = shape_factor<O, true>(s.sz_new + 1, z_new);
    const int k_old = shifted_shape_factor<O, true>(s.sz_old, z_old, k_new);
    s.dil = (i_old < i_new) ? 0 : 1; s.diu = (i_old > i_new) ? 0 : 1;
    s.djl = (j_old < j_new) ? 0 : 1; s.dju = (j_old > j_new) ? 0 : 1;
    s.dkl = (k_old < k_new) ? 0 : 1; s.dku = (k_old > k_new) ? 0 : 1;
    s.bi = g.lo0 + i_new - 1; s.bj = g.lo1 + j_new - 1; s.bk = g.lo2 + k_new - 1;
}

// Sink concept: void add(int comp, int i, int j, int k, double v) with i,j,k relative to slot 0.
//
// Loop structure: all loops run over the full static slot range so that every register
// array index is a compile-time constant (no scratch spills).  The reference trims the
// ranges with dil/diu... (CurrentDeposition.H:777-788); outside the trimmed ranges both
// shape arrays are exactly zero, so a transverse row is skipped when its weight T is zero
// and the longitudinal ends are predicated on dil/diu -- the set of non-zero deposits and
// the arithmetic of each one are those of the reference.
// One transverse row of the Esirkepov deposit: running sum along the longitudinal direction
// (CurrentDeposition.H:794-799).  Slots 1..O are inside the trimmed range for every particle;
// slot 0 / slot O+1 only for particles that moved down / up one cell, so those two are issued
// only when some lane of the wave needs them (any_lo / any_hi are wave-uniform): an LDS atomic
// with an empty exec mask would still cost an issue slot on the (binding) LDS pipe.
template <int O, int COMP, class Sink>
__device__ __forceinline__ void esirkepov_row(Sink& sink, const double (&d)[O + 2], double T, int dl, int du,
                                              bool any_lo, bool any_hi, int a, int b) {
    double sd[O + 2];
    double run = 0.;
#pragma unroll
    for (int l = 0; l <= O + 1; l++) {
        run += d[l] * T;
        sd[l] = run;
    }
    auto put = [&](int l, double v) {
        if constexpr (COMP == 0) sink.add(0, l, a, b, v);        // row (j=a, k=b), running along i
        else if constexpr (COMP == 1) sink.add(1, a, l, b, v);   // row (i=a, k=b), running along j
        else sink.add(2, a, b, l, v);                            // row (i=a, j=b), running along k
    };
#pragma unroll
    for (int l = 1; l <= O; l++) put(l, sd[l]);
    if (any_lo) { if (dl == 0) put(0, sd[0]); }
    if (any_hi) { if (du == 0) put(O + 1, sd[O + 1]); }
}

// One component (COMP = 0,1,2 -> Jx,Jy,Jz) of one particle.  Separate so that the tile kernel can
// spread the three components of its (rare) cell-crossing particles over different waves.
// The loop over the slow transverse direction b is NOT unrolled: this path runs once per tile on
// a handful of particles, and as straight-line code (~15 KB per component) it was bound by
// instruction fetch, not by arithmetic.  The rows inside one b are static as before.
template <int O, int COMP, class Sink>
__device__ __forceinline__ void esirkepov_accumulate_comp(const EsirkepovShapes<O>& s, const EsirkepovStep& es,
                                                          Sink& sink) {
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    // longitudinal direction L, transverse directions A (fast index of the row) and B
    const double* Ln = COMP == 0 ? s.sx_new : COMP == 1 ? s.sy_new : s.sz_new;
    const double* Lo = COMP == 0 ? s.sx_old : COMP == 1 ? s.sy_old : s.sz_old;
    const double* An = COMP == 0 ? s.sy_new : s.sx_new;
    const double* Ao = COMP == 0 ? s.sy_old : s.sx_old;
    const double* Bn = COMP == 2 ? s.sy_new : s.sz_new;
    const double* Bo = COMP == 2 ? s.sy_old : s.sz_old;
    const double invdtd = es.invdtd[COMP];
    const int dl = COMP == 0 ? s.dil : COMP == 1 ? s.djl : s.dkl;
    const int du = COMP == 0 ? s.diu : COMP == 1 ? s.dju : s.dku;
    double d[O + 2];
#pragma unroll
    for (int a = 0; a < O + 2; ++a) d[a] = s.wq * invdtd * sub_rn(Lo[a], Ln[a]);
    const bool lo = __builtin_amdgcn_ballot_w64(dl == 0) != 0, hi = __builtin_amdgcn_ballot_w64(du == 0) != 0;
    double an[O + 3], ao[O + 3], bnv[O + 3], bov[O + 3];   // by value: the run-time loop must not see the struct
#pragma unroll
    for (int a = 0; a < O + 3; ++a) { an[a] = An[a]; ao[a] = Ao[a]; bnv[a] = Bn[a]; bov[a] = Bo[a]; }
#pragma unroll 1
    for (int b = 0; b <= O + 2; b++) {
        // slot b sits in element 0: the arrays are shifted down once per trip, so that every register
        // index stays a compile-time constant (a run-time index would move them to scratch memory)
        const double bn = bnv[0], bo = bov[0];
#pragma unroll
        for (int i = 0; i < O + 2; ++i) { bnv[i] = bnv[i + 1]; bov[i] = bov[i + 1]; }
        if (__builtin_amdgcn_ballot_w64(bn != 0.0 || bo != 0.0) != 0) {   // some lane has weight on this plane
#pragma unroll
            for (int a = 0; a <= O + 2; a++) {
                const double T = one_third * (an[a] * bn + ao[a] * bo) + one_sixth * (an[a] * bo + ao[a] * bn);
                if (T != 0.0) esirkepov_row<O, COMP>(sink, d, T, dl, du, lo, hi, a, b);
            }
        }
    }
}

template <int O, class Sink>
__device__ __forceinline__ void esirkepov_accumulate(const EsirkepovShapes<O>& s, const EsirkepovStep& es, Sink& sink) {
    esirkepov_accumulate_comp<O, 0>(s, es, sink);
    esirkepov_accumulate_comp<O, 1>(s, es, sink);
    esirkepov_accumulate_comp<O, 2>(s, es, sink);
}

// ---- fast path: particles that stay in their cell during the step ---------------------------
// For a particle with i_old == i_new in all three directions the old and the new weights sit on
// the same O+1 slots (1..O+1 of the frame), the trimmed ranges are the static dil = diu = 1, and
// every loop bound is a compile-time constant: (O+1)^2 rows of O deposits per component, no
// masks, no ballots, no zero tests.  In a thermal plasma ~98 % of the particles qualify
// (u_th = 0.01 c moves a particle 0.006 cell per step), so the tile kernel routes them here and
// keeps the general code above for the particles with a cell crossing.
//
// Two particles of the same frame are merged before the atomics: cell-sorted neighbours share
// the slot frame (same i_new, j_new, k_new), their row values are summed in registers and
// deposited with ONE atomic per slot, which halves the load on the LDS-atomic pipe.
// Arithmetic, per component (here Jx; rows (j,k), running along i), per particle p:
//   D_p[l]   = sum_{l'<=l} wq invdtd.x (sx_old[l'] - sx_new[l'])           (running sum, once per component)
//   T_p(j,k) = sy_new[j] (1/3 sz_new[k] + 1/6 sz_old[k]) + sy_old[j] (1/3 sz_old[k] + 1/6 sz_new[k])
//   deposit(l,j,k) = D_1[l] T_1 + D_2[l] T_2
// which is the reference's sdxi (CurrentDeposition.H:794-799) with the sums re-associated: the
// transverse weight is factored through the z-dependent pair (computed once per k), and the running
// sum is taken over D instead of over D*T.  Differences are at round-off (tests: 1e-12 of max|J|).
template <int O>
struct EsirkepovNC {
    double n[3][O + 1], o[3][O + 1];   // new / old weights per direction on slots 1..O+1
    double wq;
};

// Integer cell of a grid coordinate, as Compute_shape_factor / Compute_shifted_shape_factor
// index it (ShapeFactors.H:27-156): the crossing test i_old != i_new of CurrentDeposition.H:777-788.
template <int O>
__device__ __forceinline__ int shape_cell(double x) {
    if constexpr (O % 2 == 0) return (int)(x + 0.5);   // even orders: the nearest node
    else return (int)floor(x);
}

template <int O>
__device__ __forceinline__ void esirkepov_nc_shapes(const ParticleState& p, const Geom& g, double q,
                                                    const EsirkepovStep& es, EsirkepovNC<O>& s) {
    s.wq = q * p.w;
    const EsirkepovCoords cc = esirkepov_coords(p, g, es);
    const double x_new = cc.x_new, x_old = cc.x_old, y_new = cc.y_new, y_old = cc.y_old, z_new = cc.z_new,
                 z_old = cc.z_old;
    // old weights on the node of the NEW position: the caller guarantees i_old == i_new up to the
    // rounding of x_old on a cell boundary (see shape_weights_at)
    shape_weights_at<O, true>(s.o[0], x_old, shape_node<O>(shape_factor<O, true>(s.n[0], x_new)));
    shape_weights_at<O, true>(s.o[1], y_old, shape_node<O>(shape_factor<O, true>(s.n[1], y_new)));
    shape_weights_at<O, true>(s.o[2], z_old, shape_node<O>(shape_factor<O, true>(s.n[2], z_new)));
}

// frame (slot-0 grid index) and crossing flag from the grid coordinates
template <int O>
__device__ __forceinline__ bool esirkepov_frame_cross(const EsirkepovCoords& cc, const Geom& g, int& bi, int& bj,
                                                      int& bk) {
    bi = g.lo0 + shape_node_of<O>(cc.x_new) - (O / 2 + 1);   // one slot below the leftmost stencil point (j, j - 1, j - 1, j - 2 for orders 1 .. 4)
    bj = g.lo1 + shape_node_of<O>(cc.y_new) - (O / 2 + 1);   // one slot below the leftmost stencil point (j, j - 1, j - 1, j - 2 for orders 1 .. 4)
    bk = g.lo2 + shape_node_of<O>(cc.z_new) - (O / 2 + 1);   // one slot below the leftmost stencil point (j, j - 1, j - 1, j - 2 for orders 1 .. 4)
    return shape_cell<O>(cc.x_old) != shape_cell<O>(cc.x_new) || shape_cell<O>(cc.y_old) != shape_cell<O>(cc.y_new) ||
           shape_cell<O>(cc.z_old) != shape_cell<O>(cc.z_new);
}

template <int O>
__device__ __forceinline__ bool esirkepov_frame_cross(const ParticleState& p, const Geom& g, const EsirkepovStep& es,
                                                      int& bi, int& bj, int& bk) {
    return esirkepov_frame_cross<O>(esirkepov_coords(p, g, es), g, bi, bj, bk);
}

// sink slot 0 = the frame's slot 0 (grid index bi,bj,bk)
template <int O, class Sink>
__device__ __forceinline__ void esirkepov_accumulate_pair_nc(const EsirkepovNC<O>& s1, const EsirkepovNC<O>& s2,
                                                             bool null2, const EsirkepovStep& es, Sink& sink) {
    constexpr int NW = O + 1;
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    const double (&invdtd)[3] = es.invdtd;
    const double wq2 = null2 ? 0.0 : s2.wq;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // longitudinal direction c; transverse t1 (rows' inner index) and t2 (outer index):
        // Jx: (y,z), Jy: (x,z), Jz: (x,y) -- the row orders of CurrentDeposition.H:792-824
        const int t1 = c == 0 ? 1 : 0;
        const int t2 = c == 2 ? 1 : 2;
        double D1[O], D2[O];
        {
            double r1 = 0.0, r2 = 0.0;
#pragma unroll
            for (int l = 0; l < O; ++l) {
                r1 += s1.wq * invdtd[c] * sub_rn(s1.o[c][l], s1.n[c][l]);
                r2 += wq2 * invdtd[c] * sub_rn(s2.o[c][l], s2.n[c][l]);
                D1[l] = r1; D2[l] = r2;
            }
        }
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const double p1 = one_third * s1.n[t2][b] + one_sixth * s1.o[t2][b];
            const double q1 = one_third * s1.o[t2][b] + one_sixth * s1.n[t2][b];
            const double p2 = one_third * s2.n[t2][b] + one_sixth * s2.o[t2][b];
            const double q2 = one_third * s2.o[t2][b] + one_sixth * s2.n[t2][b];
#pragma unroll
            for (int a = 0; a < NW; ++a) {
                const double T1 = s1.n[t1][a] * p1 + s1.o[t1][a] * q1;
                const double T2 = s2.n[t1][a] * p2 + s2.o[t1][a] * q2;
#pragma unroll
                for (int l = 0; l < O; ++l) {
                    const double v = D1[l] * T1 + D2[l] * T2;
                    if (c == 0) sink.add(0, l + 1, a + 1, b + 1, v);
                    else if (c == 1) sink.add(1, a + 1, l + 1, b + 1, v);
                    else sink.add(2, a + 1, b + 1, l + 1, v);
                }
            }
        }
    }
}

// The same pair deposit as esirkepov_accumulate_pair_nc (same formulas, same order of the sums), organised for a small
// register footprint: the three components run as three phases separated by scheduling barriers, each phase holds only
// the weights it needs, and the weights of a direction that two phases need (x: Jz and Jy) are evaluated again from the
// grid coordinate instead of being carried across the phase in between (RECOMPUTE_X; 4 spline evaluations per pair).
// Live state per phase at order 3: Jz 16 + 16 + 6 doubles (x, y weights of both particles, Dz), Jx 16 + 16 + 6 (+ 6 for
// Dy, + the x coordinates), Jy 16 + 16 + 6 -- against 2 x 24 weights + 2 x 9 running sums when everything is kept.
// Both particles of a pair share the stencil frame, i.e. the reference node of every direction.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// before_jx / before_jy: called between the phases (the tile kernel requests its next chunk's particles there: the
// phases that follow hide the loads' latency, and the registers of the phases that are over are free for them)
template <int O, bool RECOMPUTE_X, class Sink, class HookX = NoHook, class HookY = NoHook>
__device__ __forceinline__ void esirkepov_pair_phased(EsirkepovCoords c1, EsirkepovCoords c2, const double wq1,
                                                      const double wq2, const EsirkepovStep& es, Sink& sink,
                                                      HookX&& before_jx = NoHook{}, HookY&& before_jy = NoHook{}) {
    constexpr int NW = O + 1;
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    const int jx = shape_node_of<O>(c1.x_new), jy = shape_node_of<O>(c1.y_new), jz = shape_node_of<O>(c1.z_new);
    // running sums of wq invdtd (s_old - s_new) along one direction (CurrentDeposition.H:794-799, re-associated as in
    // esirkepov_accumulate_pair_nc)
    auto running = [&](double (&D)[O], const double wq, const double invdtd, const double xn, const double xo, const int j) {
        double n[NW], o[NW];
        bspline_weights<O, true>(n, xn, j);
        bspline_weights<O, true>(o, xo, j);
        double r = 0.0;
#pragma unroll
        for (int l = 0; l < O; ++l) {
            r += wq * invdtd * sub_rn(o[l], n[l]);
            D[l] = r;
        }
    };
    // one component: D along the longitudinal direction, (an, ao) weights of the rows' inner index, (bn, bo) of the outer
    auto phase = [&](auto comp, const double (&D1)[O], const double (&D2)[O], const double (&a1n)[NW],
                     const double (&a1o)[NW], const double (&a2n)[NW], const double (&a2o)[NW], const double (&b1n)[NW],
                     const double (&b1o)[NW], const double (&b2n)[NW], const double (&b2o)[NW]) {
        constexpr int c = decltype(comp)::value;
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const double P1 = one_third * b1n[b] + one_sixth * b1o[b];
            const double Q1 = one_third * b1o[b] + one_sixth * b1n[b];
            const double P2 = one_third * b2n[b] + one_sixth * b2o[b];
            const double Q2 = one_third * b2o[b] + one_sixth * b2n[b];
#pragma unroll
            for (int a = 0; a < NW; ++a) {
                const double T1 = a1n[a] * P1 + a1o[a] * Q1;
                const double T2 = a2n[a] * P2 + a2o[a] * Q2;
#pragma unroll
                for (int l = 0; l < O; ++l) {
                    const double v = D1[l] * T1 + D2[l] * T2;
                    if constexpr (c == 0) sink.add(0, l + 1, a + 1, b + 1, v);
                    else if constexpr (c == 1) sink.add(1, a + 1, l + 1, b + 1, v);
                    else sink.add(2, a + 1, b + 1, l + 1, v);
                }
            }
        }
    };
    double x1n[NW], x1o[NW], x2n[NW], x2o[NW];
    bspline_weights<O, true>(x1n, c1.x_new, jx); bspline_weights<O, true>(x1o, c1.x_old, jx);
    bspline_weights<O, true>(x2n, c2.x_new, jx); bspline_weights<O, true>(x2o, c2.x_old, jx);
    double y1n[NW], y1o[NW], y2n[NW], y2o[NW];
    bspline_weights<O, true>(y1n, c1.y_new, jy); bspline_weights<O, true>(y1o, c1.y_old, jy);
    bspline_weights<O, true>(y2n, c2.y_new, jy); bspline_weights<O, true>(y2o, c2.y_old, jy);
    {   // Jz: rows (i, j), running along k
        double D1[O], D2[O];
        running(D1, wq1, es.invdtd[2], c1.z_new, c1.z_old, jz);
        running(D2, wq2, es.invdtd[2], c2.z_new, c2.z_old, jz);
        phase(std::integral_constant<int, 2>{}, D1, D2, x1n, x1o, x2n, x2o, y1n, y1o, y2n, y2o);
    }
    __builtin_amdgcn_sched_barrier(0);
    before_jx();
    double z1n[NW], z1o[NW], z2n[NW], z2o[NW];
    bspline_weights<O, true>(z1n, c1.z_new, jz); bspline_weights<O, true>(z1o, c1.z_old, jz);
    bspline_weights<O, true>(z2n, c2.z_new, jz); bspline_weights<O, true>(z2o, c2.z_old, jz);
    double Dy1[O], Dy2[O];
    {   // Jx: rows (j, k), running along i; Dy is taken here, before the y weights die
        double D1[O], D2[O];
        double r1 = 0.0, r2 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int l = 0; l < O; ++l) {
            r1 += wq1 * es.invdtd[0] * sub_rn(x1o[l], x1n[l]);
            r2 += wq2 * es.invdtd[0] * sub_rn(x2o[l], x2n[l]);
            s1 += wq1 * es.invdtd[1] * sub_rn(y1o[l], y1n[l]);
            s2 += wq2 * es.invdtd[1] * sub_rn(y2o[l], y2n[l]);
            D1[l] = r1; D2[l] = r2; Dy1[l] = s1; Dy2[l] = s2;
        }
        if constexpr (RECOMPUTE_X) {
            // the x weights are not carried across this phase: make the coordinates opaque so that the second
            // evaluation below is not merged with the first one
            WXA_OPAQUE_F64(c1.x_new); WXA_OPAQUE_F64(c1.x_old); WXA_OPAQUE_F64(c2.x_new); WXA_OPAQUE_F64(c2.x_old);
        }
        phase(std::integral_constant<int, 0>{}, D1, D2, y1n, y1o, y2n, y2o, z1n, z1o, z2n, z2o);
    }
    __builtin_amdgcn_sched_barrier(0);
    before_jy();
    if constexpr (RECOMPUTE_X) {
        bspline_weights<O, true>(x1n, c1.x_new, jx); bspline_weights<O, true>(x1o, c1.x_old, jx);
        bspline_weights<O, true>(x2n, c2.x_new, jx); bspline_weights<O, true>(x2o, c2.x_old, jx);
    }
    // Jy: rows (i, k), running along j
    phase(std::integral_constant<int, 1>{}, Dy1, Dy2, x1n, x1o, x2n, x2o, z1n, z1o, z2n, z2o);
}

template <int O, bool RECOMPUTE_X, class Sink>
__device__ __forceinline__ void esirkepov_pair_phased(const ParticleState& p1, const ParticleState& p2, const bool null2,
                                                      const Geom& g, const double q, const EsirkepovStep& es,
                                                      Sink& sink) {
    esirkepov_pair_phased<O, RECOMPUTE_X>(esirkepov_coords(p1, g, es), esirkepov_coords(p2, g, es), q * p1.w,
                                          null2 ? 0.0 : q * p2.w, es, sink);
}

// One component of one particle that may cross a cell face in any direction (but stays on the tile): the Esirkepov
// sums of CurrentDeposition.H:777-824 on a frame of O+2 slots per direction that starts at the lower of the old and the
// new reference node, with compile-time loop bounds -- weights outside a position's own O+1 slots are exact zeros, so the
// extra rows and entries deposit nothing (+0.0) and the non-zero ones are the reference's.  ~2.5x the work of the
// no-crossing body instead of the ~13x of the fully general one spread over (component, plane) lanes; used for the
// 1-2 % of the particles that cross a cell in a step.  `base` returns the grid index of slot 0 per direction.
template <int O>
struct WideFrame {
    int b[3];           // grid index of slot 0 (x, y, z)
    int sn[3], so[3];   // offset (0 or 1) of the new / old weights inside the frame
    int jn[3], jo[3];   // reference nodes
};
template <int O>
__device__ __forceinline__ WideFrame<O> esirkepov_wide_frame(const EsirkepovCoords& cc, const Geom& g) {
    WideFrame<O> f;
    const double xn[3] = {cc.x_new, cc.y_new, cc.z_new}, xo[3] = {cc.x_old, cc.y_old, cc.z_old};
    const int lo[3] = {g.lo0, g.lo1, g.lo2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        f.jn[d] = shape_node_of<O>(xn[d]);
        f.jo[d] = O == 1 ? (int)floor(xo[d]) : shape_node_of<O>(xo[d]);   // ShapeFactors.H:112 floors at order 1
        const int jm = min(f.jn[d], f.jo[d]);
        f.sn[d] = f.jn[d] - jm; f.so[d] = f.jo[d] - jm;
        f.b[d] = lo[d] + jm - O / 2;   // the leftmost stencil point of the lower of the two nodes
    }
    return f;
}
template <int O, int COMP, class Sink>
__device__ __forceinline__ void esirkepov_single_wide(const EsirkepovCoords& cc, const WideFrame<O>& f, const double wq,
                                                      const EsirkepovStep& es, Sink& sink) {
    constexpr int NW = O + 2;
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    const double xn[3] = {cc.x_new, cc.y_new, cc.z_new}, xo[3] = {cc.x_old, cc.y_old, cc.z_old};
    // weights of direction d on the wide frame: the O+1 weights at offset 0 or 1, zero elsewhere
    auto wide = [&](double (&wn)[NW], double (&wo)[NW], const int d) {
        double n[O + 1], o[O + 1];
        bspline_weights<O, true>(n, xn[d], f.jn[d]);
        bspline_weights<O, true>(o, xo[d], f.jo[d]);
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const double n0 = a <= O ? n[a < O + 1 ? a : O] : 0.0, n1 = a >= 1 ? n[a - 1 < 0 ? 0 : a - 1] : 0.0;
            const double o0 = a <= O ? o[a < O + 1 ? a : O] : 0.0, o1 = a >= 1 ? o[a - 1 < 0 ? 0 : a - 1] : 0.0;
            wn[a] = f.sn[d] ? n1 : n0;
            wo[a] = f.so[d] ? o1 : o0;
        }
    };
    constexpr int dl = COMP, da = COMP == 0 ? 1 : 0, db = COMP == 2 ? 1 : 2;   // longitudinal, inner and outer transverse
    double D[O + 1];
    {
        double ln[NW], lo_[NW];
        wide(ln, lo_, dl);
        double r = 0.0;
#pragma unroll
        for (int l = 0; l <= O; ++l) {
            r += wq * es.invdtd[COMP] * sub_rn(lo_[l], ln[l]);
            D[l] = r;
        }
    }
    double an[NW], ao[NW], bn[NW], bo[NW];
    wide(an, ao, da);
    wide(bn, bo, db);
#pragma unroll
    for (int b = 0; b < NW; ++b) {
        const double P = one_third * bn[b] + one_sixth * bo[b];
        const double Q = one_third * bo[b] + one_sixth * bn[b];
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const double T = an[a] * P + ao[a] * Q;
#pragma unroll
            for (int l = 0; l <= O; ++l) {
                const double v = D[l] * T;
                if constexpr (COMP == 0) sink.add(0, l, a, b, v);
                else if constexpr (COMP == 1) sink.add(1, a, l, b, v);
                else sink.add(2, a, b, l, v);
            }
        }
    }
}

// A wide frame in four registers instead of fifteen: the new reference nodes and the six offset flags (slot 0, the old nodes
// and the offsets follow from them).  The streaming loop of the tile kernel keeps its two particles' frames in this form
// and unpacks them component by component.
struct PackedWideFrame {
    int jn[3];
    int bits;   // bit d: sn[d], bit 3 + d: so[d]
};
template <int O>
__device__ __forceinline__ PackedWideFrame pack_wide_frame(const WideFrame<O>& f) {
    PackedWideFrame p;
    p.bits = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        p.jn[d] = f.jn[d];
        p.bits |= (f.sn[d] << d) | (f.so[d] << (3 + d));
    }
    return p;
}
template <int O>
__device__ __forceinline__ WideFrame<O> unpack_wide_frame(const PackedWideFrame& p, const Geom& g) {
    WideFrame<O> f;
    const int lo[3] = {g.lo0, g.lo1, g.lo2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        f.jn[d] = p.jn[d];
        f.sn[d] = (p.bits >> d) & 1; f.so[d] = (p.bits >> (3 + d)) & 1;
        const int jm = f.jn[d] - f.sn[d];
        f.jo[d] = jm + f.so[d];
        f.b[d] = lo[d] + jm - O / 2;   // the leftmost stencil point of the lower of the two nodes
    }
    return f;
}

// esirkepov_single_wide for two particles whose wide frames start at the same point (both frames' slot 0; the offsets of the
// old and new weights inside the frame are each particle's own): every point receives the sum of the two values, half the
// LDS atomics per particle.  wq2 = 0: particle 1 alone (particle 2's weights are computed and multiplied away).
template <int O, int COMP, class Sink>
__device__ __forceinline__ void esirkepov_pair_wide(const EsirkepovCoords& c1, const WideFrame<O>& f1, const double wq1,
                                                    const EsirkepovCoords& c2, const WideFrame<O>& f2, const double wq2,
                                                    const EsirkepovStep& es, Sink& sink) {
    constexpr int NW = O + 2;
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    auto wide = [&](const EsirkepovCoords& cc, const WideFrame<O>& f, double (&wn)[NW], double (&wo)[NW], const int d) {
        const double xn = d == 0 ? cc.x_new : d == 1 ? cc.y_new : cc.z_new, xo = d == 0 ? cc.x_old : d == 1 ? cc.y_old : cc.z_old;
        double n[O + 1], o[O + 1];
        bspline_weights<O, true>(n, xn, f.jn[d]);
        bspline_weights<O, true>(o, xo, f.jo[d]);
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const double n0 = a <= O ? n[a < O + 1 ? a : O] : 0.0, n1 = a >= 1 ? n[a - 1 < 0 ? 0 : a - 1] : 0.0;
            const double o0 = a <= O ? o[a < O + 1 ? a : O] : 0.0, o1 = a >= 1 ? o[a - 1 < 0 ? 0 : a - 1] : 0.0;
            wn[a] = f.sn[d] ? n1 : n0;
            wo[a] = f.so[d] ? o1 : o0;
        }
    };
    constexpr int dl = COMP, da = COMP == 0 ? 1 : 0, db = COMP == 2 ? 1 : 2;   // longitudinal, inner and outer transverse
    double D1[O + 1], D2[O + 1];
    auto running = [&](const EsirkepovCoords& cc, const WideFrame<O>& f, const double wq, double (&D)[O + 1]) {
        double ln[NW], lo_[NW];
        wide(cc, f, ln, lo_, dl);
        double r = 0.0;
#pragma unroll
        for (int l = 0; l <= O; ++l) {
            r += wq * es.invdtd[COMP] * sub_rn(lo_[l], ln[l]);
            D[l] = r;
        }
    };
    running(c1, f1, wq1, D1);
    running(c2, f2, wq2, D2);
    double a1n[NW], a1o[NW], b1n[NW], b1o[NW], a2n[NW], a2o[NW], b2n[NW], b2o[NW];
    wide(c1, f1, a1n, a1o, da); wide(c1, f1, b1n, b1o, db);
    wide(c2, f2, a2n, a2o, da); wide(c2, f2, b2n, b2o, db);
#pragma unroll
    for (int b = 0; b < NW; ++b) {
        const double P1 = one_third * b1n[b] + one_sixth * b1o[b], Q1 = one_third * b1o[b] + one_sixth * b1n[b];
        const double P2 = one_third * b2n[b] + one_sixth * b2o[b], Q2 = one_third * b2o[b] + one_sixth * b2n[b];
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const double T1 = a1n[a] * P1 + a1o[a] * Q1, T2 = a2n[a] * P2 + a2o[a] * Q2;
#pragma unroll
            for (int l = 0; l <= O; ++l) {
                const double v = D1[l] * T1 + D2[l] * T2;
                if constexpr (COMP == 0) sink.add(0, l, a, b, v);
                else if constexpr (COMP == 1) sink.add(1, a, l, b, v);
                else sink.add(2, a, b, l, v);
            }
        }
    }
}

// The sum of v over the wave's 64 lanes, valid in lanes 48 .. 63: six DPP steps (lane ^ 1, lane ^ 2, mirror of 8, mirror of
// 16 -- every lane of a row of 16 then holds its row's sum -- lane 15 of the row before, lane 31), i.e. twelve v_mov_dpp
// and six v_add_f64 on the VALU, nothing on the LDS pipe.  Rows 0 .. 2 end with partial sums (the broadcasts go to every
// row, so no `old` operand has to be zeroed).  For waves whose lanes all deposit on the SAME points (the cells of a wake's
// density spike hold 10^3 .. 10^5 particles): a ds_add_f64 whose 64 lanes share an address is served lane by lane, on the
// one LDS pipe of the CU.
#ifndef WXA_HAVE_WAVE_SUM_F64   // tests/hipcpu: wave shuffles
__device__ __forceinline__ double wave_sum_f64(double v) {
#define WXA_DPP_ADD(ctrl)                                                                      \
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true),   \
                          __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true))
    WXA_DPP_ADD(0xB1);    // quad_perm [1, 0, 3, 2]
    WXA_DPP_ADD(0x4E);    // quad_perm [2, 3, 0, 1]
    WXA_DPP_ADD(0x141);   // row_half_mirror
    WXA_DPP_ADD(0x140);   // row_mirror
    WXA_DPP_ADD(0x142);   // row_bcast:15: row r += the sum of row r - 1
    WXA_DPP_ADD(0x143);   // row_bcast:31: rows 2, 3 += rows 0 + 1 (lane 31)
#undef WXA_DPP_ADD
    return v;
}
#endif
// Lanes l and l + 32 of a wave: the other lane's value of a wave-uniformly executed expression (v_permlane32_swap: the
// upper 32 lanes of its first operand change places with the lower 32 of its second).
#ifndef WXA_HAVE_PARTNER32   // tests/hipcpu: wave shuffles
__device__ __forceinline__ int partner32(const int v) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ double partner32_f64(const double v) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return (threadIdx.x & 32) ? __hiloint2double((int)hi[0], (int)lo[0]) : __hiloint2double((int)hi[1], (int)lo[1]);
}
#endif
// LdsSink for lanes l and l + 32 of the chunk layout -- two pairs of ONE cell -- in the streaming deposition: where the two
// lanes deposit on the same wide frame (`shared`), every value is summed over the two and the lower lane adds it; elsewhere
// each lane adds its own.  Half the LDS atomics of a shared frame, and none of the same-address conflicts that two lanes of
// one instruction on one frame cost (37 % of the LDS-active cycles of BASELINE config 5, profiles/round5).  Executed by
// every lane of the wave (the exchange is a wave operation): a lane without a particle takes part with zero weight.
template <class Sink>
struct PairSumSink {
    Sink inner;
    bool shared, adds;   // adds: this lane issues the atomic (its own frame, or the pair's as the lower lane)
    __device__ __forceinline__ PairSumSink(const Sink& s, bool shared_, bool adds_) : inner(s), shared(shared_), adds(adds_) {}
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        const double t = partner32_f64(v);
        const double s = shared ? v + t : v;
        if (adds) inner.add(c, i, j, k, s);
    }
};

// Two values per lane in, one sum per lane out: lane l < 32 gets a(l) + a(l + 32), lane l + 32 gets b(l) + b(l + 32) -- one
// v_permlane32_swap per dword (the upper 32 lanes of a change places with the lower 32 of b) and one v_add_f64.
#ifndef WXA_HAVE_SWAP_ADD_HALVES   // tests/hipcpu: wave shuffles
__device__ __forceinline__ double swap_add_halves(const double a, const double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
#endif
// PairSumSink with the adding shared out as well, for a wave whose lane pairs (l, l + 32) ALL deposit on one frame per pair:
// the body's values arrive along the component's own direction (the running sums D[0 .. O], O odd: an even number), two
// neighbouring points at a time; the pair's sums of the first go to the lower lane, of the second to the upper lane
// (swap_add_halves), and ONE ds_add_f64 of the wave adds both -- the upper lane's sink starts one point further along that
// direction.  Half the LDS-atomic instructions of PairSumSink, which is what the LDS charges for (an atomic instruction
// costs the same with 32 or 64 active lanes, profiles/round6/README.md session j).
template <class Sink>
struct PairScatterSink {
    Sink inner;
    bool active;   // the lane pair has a frame (a lane without a particle joins its partner's with zero weights)
    double held = 0.0;
    int hi = 0, hj = 0, hk = 0;
    bool full = false;
    __device__ __forceinline__ PairScatterSink(const Sink& s, bool active_) : inner(s), active(active_) {}
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        if (!full) {
            held = v; hi = i; hj = j; hk = k; full = true;
            return;
        }
        full = false;
        const double s = swap_add_halves(held, v);
        if (active) inner.add(c, hi, hj, hk, s);
    }
};

// LdsSink for a wave whose lanes share the frame: every value is summed over the wave, lane 63 adds it
template <class Sink>
struct WaveSumSink {
    Sink inner;
    bool last;
    __device__ __forceinline__ WaveSumSink(const Sink& s, bool is_last_lane) : inner(s), last(is_last_lane) {}
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        const double s = wave_sum_f64(v);
        if (last) inner.add(c, i, j, k, s);
        // one value after the other: left to itself the scheduler interleaves all the frame's sums and spills
        __builtin_amdgcn_sched_barrier(0);
    }
};

// One component of one particle that stays in its cell, on its own fast frame (slot 0 = the frame's first point, weights on
// slots 1 .. O+1, like the pair body above with an empty partner): (O+1)^2 rows of O deposits.  Phase D of the tile kernel
// for particles that could not be merged with their lane partner.
template <int O, int COMP, class Sink>
__device__ __forceinline__ void esirkepov_single_fast(const EsirkepovCoords& cc, const double wq, const EsirkepovStep& es,
                                                      Sink& sink) {
    constexpr int NW = O + 1;
    constexpr double one_third = 1.0 / 3.0, one_sixth = 1.0 / 6.0;
    const double xn[3] = {cc.x_new, cc.y_new, cc.z_new}, xo[3] = {cc.x_old, cc.y_old, cc.z_old};
    constexpr int dl = COMP, da = COMP == 0 ? 1 : 0, db = COMP == 2 ? 1 : 2;   // longitudinal, inner and outer transverse
    const int jl = shape_node_of<O>(xn[dl]), ja = shape_node_of<O>(xn[da]), jb = shape_node_of<O>(xn[db]);
    double D[O];
    {
        double n[NW], o[NW];
        bspline_weights<O, true>(n, xn[dl], jl);
        bspline_weights<O, true>(o, xo[dl], jl);
        double r = 0.0;
#pragma unroll
        for (int l = 0; l < O; ++l) {
            r += wq * es.invdtd[COMP] * sub_rn(o[l], n[l]);
            D[l] = r;
        }
    }
    double an[NW], ao[NW], bn[NW], bo[NW];
    bspline_weights<O, true>(an, xn[da], ja); bspline_weights<O, true>(ao, xo[da], ja);
    bspline_weights<O, true>(bn, xn[db], jb); bspline_weights<O, true>(bo, xo[db], jb);
#pragma unroll
    for (int b = 0; b < NW; ++b) {
        const double P = one_third * bn[b] + one_sixth * bo[b];
        const double Q = one_third * bo[b] + one_sixth * bn[b];
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const double T = an[a] * P + ao[a] * Q;
#pragma unroll
            for (int l = 0; l < O; ++l) {
                const double v = D[l] * T;
                if constexpr (COMP == 0) sink.add(0, l + 1, a + 1, b + 1, v);
                else if constexpr (COMP == 1) sink.add(1, a + 1, l + 1, b + 1, v);
                else sink.add(2, a + 1, b + 1, l + 1, v);
            }
        }
    }
}

// Direct deposition on the Yee grid: jx(c,n,n) jy(n,c,n) jz(n,n,c).
template <int O>
struct DirectShapes {
    double sxn[O + 1], sxc[O + 1], syn[O + 1], syc[O + 1], szn[O + 1], szc[O + 1];
    int jn, jc, kn, kc, ln, lc;   // grid indices of the leftmost point per centring
    double wqx, wqy, wqz;
};

template <int O>
__device__ __forceinline__ void direct_shapes(const ParticleState& p, const Geom& g, double q,
                                              double relative_time, DirectShapes<O>& s) {
    const double invvol = g.dxi * g.dyi * g.dzi;
    const double clightsq = 1.0 / PhysConst::c / PhysConst::c;
    const double gaminv =
        inv_sqrt(1.0 + p.ux * p.ux * clightsq + p.uy * p.uy * clightsq + p.uz * p.uz * clightsq);
    const double vx = p.ux * gaminv, vy = p.uy * gaminv, vz = p.uz * gaminv;
    const double wq = q * p.w;
    s.wqx = wq * invvol * vx; s.wqy = wq * invvol * vy; s.wqz = wq * invvol * vz;
    const double xmid = ((p.x - g.xmin) + relative_time * vx) * g.dxi;
    const double ymid = ((p.y - g.ymin) + relative_time * vy) * g.dyi;
    const double zmid = ((p.z - g.zmin) + relative_time * vz) * g.dzi;
    s.jn = g.lo0 + shape_factor<O>(s.sxn, xmid); s.jc = g.lo0 + shape_factor<O>(s.sxc, xmid - 0.5);
    s.kn = g.lo1 + shape_factor<O>(s.syn, ymid); s.kc = g.lo1 + shape_factor<O>(s.syc, ymid - 0.5);
    s.ln = g.lo2 + shape_factor<O>(s.szn, zmid); s.lc = g.lo2 + shape_factor<O>(s.szc, zmid - 0.5);
}

// doDepositionShapeN (CurrentDeposition.H:48-249) for the two particles of a lane at once (tile kernels), one component.
// Two particles of one sort cell have the same nodal stencils; their cell-centred stencil starts at jn - 1 or at jn (the
// half of the cell the particle sits in), so a frame of O + 2 slots along the component's own direction and O + 1 along the
// others holds both: (O + 2) (O + 1)^2 sums of two products per component instead of 2 (O + 1)^3 products, 240 LDS atomics
// per pair instead of 384 at order 3.  The sink's origin is the frame's first point: (jn - 1, kn, ln) for jx, (jn, kn - 1,
// ln) for jy, (jn, kn, ln - 1) for jz, of whichever particle carries weight (a particle without weight contributes exact
// zeros whatever its stencil).
template <int O, int COMP, class Sink>
__device__ __forceinline__ void direct_pair_component(const DirectShapes<O>& a, const DirectShapes<O>& b, Sink& sink) {
    constexpr int NW = O + 1;
    const double* wa = COMP == 0 ? a.sxc : COMP == 1 ? a.syc : a.szc;
    const double* wb = COMP == 0 ? b.sxc : COMP == 1 ? b.syc : b.szc;
    const double* ta1 = COMP == 0 ? a.syn : a.sxn;
    const double* tb1 = COMP == 0 ? b.syn : b.sxn;
    const double* ta2 = COMP == 2 ? a.syn : a.szn;
    const double* tb2 = COMP == 2 ? b.syn : b.szn;
    const double wqa = COMP == 0 ? a.wqx : COMP == 1 ? a.wqy : a.wqz;
    const double wqb = COMP == 0 ? b.wqx : COMP == 1 ? b.wqy : b.wqz;
    // first slot of the particle's own cell-centred stencil in the frame: 0 or 1
    const bool offa = (COMP == 0 ? a.jc - a.jn : COMP == 1 ? a.kc - a.kn : a.lc - a.ln) != -1;
    const bool offb = (COMP == 0 ? b.jc - b.jn : COMP == 1 ? b.kc - b.kn : b.lc - b.ln) != -1;
    double fa[NW + 1], fb[NW + 1];
#pragma unroll
    for (int m = 0; m <= NW; ++m) {
        fa[m] = offa ? (m >= 1 ? wa[m >= 1 ? m - 1 : 0] : 0.0) : (m < NW ? wa[m < NW ? m : 0] : 0.0);
        fb[m] = offb ? (m >= 1 ? wb[m >= 1 ? m - 1 : 0] : 0.0) : (m < NW ? wb[m < NW ? m : 0] : 0.0);
    }
#pragma unroll
    for (int i2 = 0; i2 < NW; ++i2)
#pragma unroll
        for (int i1 = 0; i1 < NW; ++i1) {
            // COMP 0: (i1, i2) = (y, z); COMP 1: (x, z); COMP 2: (x, y)
            const double Ta = ta1[i1] * ta2[i2] * wqa;
            const double Tb = tb1[i1] * tb2[i2] * wqb;
#pragma unroll
            for (int m = 0; m <= NW; ++m) {
                const double v = fa[m] * Ta + fb[m] * Tb;
                if constexpr (COMP == 0) sink.add(0, m, i1, i2, v);
                else if constexpr (COMP == 1) sink.add(1, i1, m, i2, v);
                else sink.add(2, i1, i2, m, v);
            }
        }
}

// ... and one component of one particle on its own stencil (the tile kernels' deferred particles), the reference's products
template <int O, int COMP, class Sink>
__device__ __forceinline__ void direct_single_component(const DirectShapes<O>& s, Sink& sink) {
#pragma unroll
    for (int iz = 0; iz <= O; iz++)
#pragma unroll
        for (int iy = 0; iy <= O; iy++)
#pragma unroll
            for (int ix = 0; ix <= O; ix++) {
                if constexpr (COMP == 0) sink.add(0, ix, iy, iz, s.sxc[ix] * s.syn[iy] * s.szn[iz] * s.wqx);
                else if constexpr (COMP == 1) sink.add(1, ix, iy, iz, s.sxn[ix] * s.syc[iy] * s.szn[iz] * s.wqy);
                else sink.add(2, ix, iy, iz, s.sxn[ix] * s.syn[iy] * s.szc[iz] * s.wqz);
            }
}

// Sink concept for direct: void add_abs(int comp, int gi, int gj, int gk, double v) with
// absolute grid indices.
template <int O, class Sink>
__device__ __forceinline__ void direct_accumulate(const DirectShapes<O>& s, Sink& sink) {
#pragma unroll
    for (int iz = 0; iz <= O; iz++)
#pragma unroll
        for (int iy = 0; iy <= O; iy++)
#pragma unroll
            for (int ix = 0; ix <= O; ix++) {
                sink.add_abs(0, s.jc + ix, s.kn + iy, s.ln + iz, s.sxc[ix] * s.syn[iy] * s.szn[iz] * s.wqx);
                sink.add_abs(1, s.jn + ix, s.kc + iy, s.ln + iz, s.sxn[ix] * s.syc[iy] * s.szn[iz] * s.wqy);
                sink.add_abs(2, s.jn + ix, s.kn + iy, s.lc + iz, s.sxn[ix] * s.syn[iy] * s.szc[iz] * s.wqz);
            }
}

// Global sink: hardware fp64 atomics straight into J.
struct GlobalSink {
    double* __restrict__ base[3];
    long js[3], ks[3];
    int bi, bj, bk;          // slot-0 grid index for the relative form
    int lo[3][3];            // array lower bounds per component
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        add_abs(c, bi + i, bj + j, bk + k, v);
    }
    __device__ __forceinline__ void add_abs(int c, int gi, int gj, int gk, double v) {
        atomic_add_f64(base[c] + (long)(gi - lo[c][0]) + (long)(gj - lo[c][1]) * js[c] +
                           (long)(gk - lo[c][2]) * ks[c], v);
    }
};

__device__ __forceinline__ GlobalSink make_global_sink(const DevF& Jx, const DevF& Jy, const DevF& Jz) {
    GlobalSink s;
    s.base[0] = Jx.p; s.base[1] = Jy.p; s.base[2] = Jz.p;
    s.js[0] = Jx.js; s.js[1] = Jy.js; s.js[2] = Jz.js;
    s.ks[0] = Jx.ks; s.ks[1] = Jy.ks; s.ks[2] = Jz.ks;
    s.lo[0][0] = Jx.lo0; s.lo[0][1] = Jx.lo1; s.lo[0][2] = Jx.lo2;
    s.lo[1][0] = Jy.lo0; s.lo[1][1] = Jy.lo1; s.lo[1][2] = Jy.lo2;
    s.lo[2][0] = Jz.lo0; s.lo[2][1] = Jz.lo1; s.lo[2][2] = Jz.lo2;
    s.bi = s.bj = s.bk = 0;
    return s;
}

}  // namespace wxa
#endif
