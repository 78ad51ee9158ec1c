// FDTD Yee field kernels and guard-cell kernels for gfx950 (MI355X).
//
// EvolveB / EvolveE are pure HBM-bound stencils (0.17 flop/B): 72 B/cell and
// 96 B/cell of algorithmic traffic.  Design:
//  * one lane per i (the contiguous direction) -> every wave load is one
//    coalesced 512-B line segment; TJ rows per workgroup so that the j+-1
//    neighbour rows are served by the same CU's L1;
//  * each lane marches KC planes in k and carries the k-neighbour in registers, so
//    each array is read once per tile (+1/KC plane halo);
//  * workgroup -> tile mapping is XCD-aware (contiguous k-slabs per XCD L2);
//  * arithmetic keeps the reference's operation order and is compiled with
//    -ffp-contract=off: results are bit-identical to the CPU path.
#include "common.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace wxa {

// Tile shape: a workgroup is TW waves side by side along i (64 TW lanes: one contiguous segment of 512 TW bytes
// per row and array) x TJ rows, and every lane marches KC planes in k.  Round 1 measured TW = 1 at 256^3
// (EvolveB / EvolveE, ms back to back, same box):
//   TJ x KC = 4 x 16: 0.247 / 0.323    4 x 4: 0.233 / 0.305    2 x 4: 0.228 / 0.311
//   2 x 2: 0.233 / 0.308    1 x 4: 0.231 / 0.312    8 x 4: 0.247 / 0.310    4 x 32: 0.295 / 0.378
// Short marches win: the k-halo re-read of a 4-plane tile is served by the XCD's L2 (the tile order
// keeps k-neighbours on one XCD), and 4x more workgroups keep more loads in flight.
// NT: the arrays that are read once and written once per call (B in EvolveB; E and J in EvolveE) move with
// non-temporal loads / stores, so that they do not evict the rows of the other operand that the j / k neighbours
// re-read from L2.  WXA_STENCIL_VARIANT=<n> selects a configuration per launch (scripts/stencil_variants.py).
// BATCH: every load of the lane's KC planes is issued before the first store.  gfx950 counts loads and stores in one
// vmcnt, and the three read-modify-writes of a plane sit in three conditional blocks: as written plane by plane the ISA is
// load B -> s_waitcnt vmcnt(0) -> store B, three times per plane, each wait also draining the store before it -- one
// round trip to memory at a time per wave, latency hidden by occupancy alone.  Same arithmetic per point: bit-identical.
template <int TW_, int TJ_, int KC_, int NT_, int BATCH_ = 0>
struct StencilCfg {
    static constexpr int TW = TW_, TI = 64 * TW_, TJ = TJ_, KC = KC_, NT = NT_, BATCH = BATCH_;
};

struct TileGrid {
    int nti, ntj, ntk;
    long ntiles;
};

template <class CFG>
static TileGrid make_tiles_cfg(const Box3& ub) {
    TileGrid t;
    t.nti = (ub.hi[0] - ub.lo[0] + CFG::TI - 1) / CFG::TI;
    t.ntj = (ub.hi[1] - ub.lo[1] + CFG::TJ - 1) / CFG::TJ;
    t.ntk = (ub.hi[2] - ub.lo[2] + CFG::KC - 1) / CFG::KC;
    t.ntiles = (long)t.nti * t.ntj * t.ntk;
    return t;
}

__device__ inline bool in_ij(const Box3& b, int i, int j) {
    return i >= b.lo[0] && i < b.hi[0] && j >= b.lo[1] && j < b.hi[1];
}

template <int NT>
__device__ __forceinline__ double ld_shared(const double* p) {   // operands that neighbouring tiles read again
    if constexpr (NT >= 2) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT>
__device__ __forceinline__ double ld_once(const double* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT>
__device__ __forceinline__ void st_once(double* p, double v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Source/FieldSolver/FiniteDifferenceSolver/EvolveB.cpp:164-186 (three fused lambdas)
template <class CFG>
__global__ void __launch_bounds__(CFG::TI* CFG::TJ)
evolve_b_kernel(DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, Box3 ub, Box3 bbx, Box3 bby,
                Box3 bbz, TileGrid tg, double dt, double idx, double idy, double idz) {
    constexpr int KC = CFG::KC, NT = CFG::NT;
    const long tile = xcd_tile_id(blockIdx.x, tg.ntiles);
    if (tile >= tg.ntiles) return;
    const int ti = (int)(tile % tg.nti);
    const int tj = (int)((tile / tg.nti) % tg.ntj);
    const int tk = (int)(tile / ((long)tg.nti * tg.ntj));
    const int i = ub.lo[0] + ti * CFG::TI + (int)threadIdx.x;
    const int j = ub.lo[1] + tj * CFG::TJ + (int)threadIdx.y;
    if (i >= ub.hi[0] || j >= ub.hi[1]) return;
    const int k0 = ub.lo[2] + tk * KC;
    const int k1 = min(k0 + KC, ub.hi[2]);
    const bool px = in_ij(bbx, i, j), py = in_ij(bby, i, j), pz = in_ij(bbz, i, j);

    const double* __restrict__ ex = Ex.p + Ex.off(i, j, k0);
    const double* __restrict__ ey = Ey.p + Ey.off(i, j, k0);
    const double* __restrict__ ez = Ez.p + Ez.off(i, j, k0);
    double* __restrict__ bx = Bx.p + Bx.off(i, j, k0);
    double* __restrict__ by = By.p + By.off(i, j, k0);
    double* __restrict__ bz = Bz.p + Bz.off(i, j, k0);

    if constexpr (CFG::BATCH != 0) {
        double exv[KC + 1], eyv[KC + 1], ezc[KC], ezj[KC], ezi[KC], exj[KC], eyi[KC], b0[KC], b1[KC], b2[KC];
        bool d0[KC], d1[KC], d2[KC];
        exv[0] = ld_shared<NT>(ex); eyv[0] = ld_shared<NT>(ey);
#pragma unroll
        for (int n = 0; n < KC; ++n) {
            const int k = k0 + n;
            const bool in = k < k1;
            d0[n] = in && px && k >= bbx.lo[2] && k < bbx.hi[2];
            d1[n] = in && py && k >= bby.lo[2] && k < bby.hi[2];
            d2[n] = in && pz && k >= bbz.lo[2] && k < bbz.hi[2];
            exv[n + 1] = eyv[n + 1] = ezc[n] = ezj[n] = ezi[n] = exj[n] = eyi[n] = b0[n] = b1[n] = b2[n] = 0.0;
            if (in) {
                exv[n + 1] = ld_shared<NT>(ex + (n + 1) * Ex.ks); eyv[n + 1] = ld_shared<NT>(ey + (n + 1) * Ey.ks);
                ezc[n] = ld_shared<NT>(ez + n * Ez.ks); ezj[n] = ld_shared<NT>(ez + n * Ez.ks + Ez.js);
                ezi[n] = ld_shared<NT>(ez + n * Ez.ks + 1);
                exj[n] = ld_shared<NT>(ex + n * Ex.ks + Ex.js); eyi[n] = ld_shared<NT>(ey + n * Ey.ks + 1);
            }
            if (d0[n]) b0[n] = ld_once<NT>(bx + n * Bx.ks);
            if (d1[n]) b1[n] = ld_once<NT>(by + n * By.ks);
            if (d2[n]) b2[n] = ld_once<NT>(bz + n * Bz.ks);
        }
        // every loaded value is "used" here, in straight-line code: the compiler's wait for the loads lands in front of
        // the stores once, not inside each conditional store block (where, with stores pending, it would be vmcnt(0))
        asm volatile("" : "+v"(exv[0]), "+v"(eyv[0]));
#pragma unroll
        for (int n = 0; n < KC; ++n)
            asm volatile("" : "+v"(exv[n + 1]), "+v"(eyv[n + 1]), "+v"(ezc[n]), "+v"(ezj[n]), "+v"(ezi[n]), "+v"(exj[n]),
                              "+v"(eyi[n]), "+v"(b0[n]), "+v"(b1[n]), "+v"(b2[n]));
#pragma unroll
        for (int n = 0; n < KC; ++n) {
            if (d0[n]) st_once<NT>(bx + n * Bx.ks, b0[n] + (dt * (idz * (eyv[n + 1] - eyv[n])) - dt * (idy * (ezj[n] - ezc[n]))));
            if (d1[n]) st_once<NT>(by + n * By.ks, b1[n] + (dt * (idx * (ezi[n] - ezc[n])) - dt * (idz * (exv[n + 1] - exv[n]))));
            if (d2[n]) st_once<NT>(bz + n * Bz.ks, b2[n] + (dt * (idy * (exj[n] - exv[n])) - dt * (idx * (eyi[n] - eyv[n]))));
        }
        return;
    }
    double ex_k = ld_shared<NT>(ex), ey_k = ld_shared<NT>(ey);
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const double ex_k1 = ld_shared<NT>(ex + Ex.ks), ey_k1 = ld_shared<NT>(ey + Ey.ks);
        const double ez_c = ld_shared<NT>(ez), ez_j1 = ld_shared<NT>(ez + Ez.js), ez_i1 = ld_shared<NT>(ez + 1);
        const double ex_j1 = ld_shared<NT>(ex + Ex.js), ey_i1 = ld_shared<NT>(ey + 1);
        if (px && k >= bbx.lo[2] && k < bbx.hi[2])
            st_once<NT>(bx, ld_once<NT>(bx) + (dt * (idz * (ey_k1 - ey_k)) - dt * (idy * (ez_j1 - ez_c))));
        if (py && k >= bby.lo[2] && k < bby.hi[2])
            st_once<NT>(by, ld_once<NT>(by) + (dt * (idx * (ez_i1 - ez_c)) - dt * (idz * (ex_k1 - ex_k))));
        if (pz && k >= bbz.lo[2] && k < bbz.hi[2])
            st_once<NT>(bz, ld_once<NT>(bz) + (dt * (idy * (ex_j1 - ex_k)) - dt * (idx * (ey_i1 - ey_k))));
        ex_k = ex_k1; ey_k = ey_k1;
        ex += Ex.ks; ey += Ey.ks; ez += Ez.ks; bx += Bx.ks; by += By.ks; bz += Bz.ks;
    }
}

// Source/FieldSolver/FiniteDifferenceSolver/EvolveE.cpp:179-216
template <class CFG>
__global__ void __launch_bounds__(CFG::TI* CFG::TJ)
evolve_e_kernel(DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, DevF Jx, DevF Jy, DevF Jz,
                Box3 ub, Box3 bex, Box3 bey, Box3 bez, TileGrid tg, double dt, double idx, double idy,
                double idz) {
    constexpr int KC = CFG::KC, NT = CFG::NT;
    const long tile = xcd_tile_id(blockIdx.x, tg.ntiles);
    if (tile >= tg.ntiles) return;
    const int ti = (int)(tile % tg.nti);
    const int tj = (int)((tile / tg.nti) % tg.ntj);
    const int tk = (int)(tile / ((long)tg.nti * tg.ntj));
    const int i = ub.lo[0] + ti * CFG::TI + (int)threadIdx.x;
    const int j = ub.lo[1] + tj * CFG::TJ + (int)threadIdx.y;
    if (i >= ub.hi[0] || j >= ub.hi[1]) return;
    const int k0 = ub.lo[2] + tk * KC;
    const int k1 = min(k0 + KC, ub.hi[2]);
    const bool px = in_ij(bex, i, j), py = in_ij(bey, i, j), pz = in_ij(bez, i, j);
    constexpr double c2 = PhysConst::c * PhysConst::c;
    constexpr double mu0 = PhysConst::mu0;

    double* __restrict__ ex = Ex.p + Ex.off(i, j, k0);
    double* __restrict__ ey = Ey.p + Ey.off(i, j, k0);
    double* __restrict__ ez = Ez.p + Ez.off(i, j, k0);
    const double* __restrict__ bx = Bx.p + Bx.off(i, j, k0);
    const double* __restrict__ by = By.p + By.off(i, j, k0);
    const double* __restrict__ bz = Bz.p + Bz.off(i, j, k0);
    const double* __restrict__ jx = Jx.p + Jx.off(i, j, k0);
    const double* __restrict__ jy = Jy.p + Jy.off(i, j, k0);
    const double* __restrict__ jz = Jz.p + Jz.off(i, j, k0);

    if constexpr (CFG::BATCH != 0) {
        double bxv[KC + 1], byv[KC + 1], bzc[KC], bzj[KC], bzi[KC], bxj[KC], byi[KC], e0[KC], e1[KC], e2[KC], j0[KC], j1[KC], j2[KC];
        bool d0[KC], d1[KC], d2[KC];
        bxv[0] = ld_shared<NT>(bx - Bx.ks); byv[0] = ld_shared<NT>(by - By.ks);   // the plane below the first one
#pragma unroll
        for (int n = 0; n < KC; ++n) {
            const int k = k0 + n;
            const bool in = k < k1;
            d0[n] = in && px && k >= bex.lo[2] && k < bex.hi[2];
            d1[n] = in && py && k >= bey.lo[2] && k < bey.hi[2];
            d2[n] = in && pz && k >= bez.lo[2] && k < bez.hi[2];
            bxv[n + 1] = byv[n + 1] = bzc[n] = bzj[n] = bzi[n] = bxj[n] = byi[n] = 0.0;
            e0[n] = e1[n] = e2[n] = j0[n] = j1[n] = j2[n] = 0.0;
            if (in) {
                bxv[n + 1] = ld_shared<NT>(bx + n * Bx.ks); byv[n + 1] = ld_shared<NT>(by + n * By.ks);
                bzc[n] = ld_shared<NT>(bz + n * Bz.ks);
                bzj[n] = ld_shared<NT>(bz + n * Bz.ks - Bz.js); bzi[n] = ld_shared<NT>(bz + n * Bz.ks - 1);
                bxj[n] = ld_shared<NT>(bx + n * Bx.ks - Bx.js); byi[n] = ld_shared<NT>(by + n * By.ks - 1);
            }
            if (d0[n]) { e0[n] = ld_once<NT>(ex + n * Ex.ks); j0[n] = ld_once<NT>(jx + n * Jx.ks); }
            if (d1[n]) { e1[n] = ld_once<NT>(ey + n * Ey.ks); j1[n] = ld_once<NT>(jy + n * Jy.ks); }
            if (d2[n]) { e2[n] = ld_once<NT>(ez + n * Ez.ks); j2[n] = ld_once<NT>(jz + n * Jz.ks); }
        }
        asm volatile("" : "+v"(bxv[0]), "+v"(byv[0]));   // as in evolve_b_kernel: one wait for the loads, in front of the stores
#pragma unroll
        for (int n = 0; n < KC; ++n)
            asm volatile("" : "+v"(bxv[n + 1]), "+v"(byv[n + 1]), "+v"(bzc[n]), "+v"(bzj[n]), "+v"(bzi[n]), "+v"(bxj[n]),
                              "+v"(byi[n]), "+v"(e0[n]), "+v"(e1[n]), "+v"(e2[n]), "+v"(j0[n]), "+v"(j1[n]), "+v"(j2[n]));
#pragma unroll
        for (int n = 0; n < KC; ++n) {
            if (d0[n]) st_once<NT>(ex + n * Ex.ks, e0[n] + c2 * dt * (-(idz * (byv[n + 1] - byv[n])) + (idy * (bzc[n] - bzj[n])) - mu0 * j0[n]));
            if (d1[n]) st_once<NT>(ey + n * Ey.ks, e1[n] + c2 * dt * (-(idx * (bzc[n] - bzi[n])) + (idz * (bxv[n + 1] - bxv[n])) - mu0 * j1[n]));
            if (d2[n]) st_once<NT>(ez + n * Ez.ks, e2[n] + c2 * dt * (-(idy * (bxv[n + 1] - bxj[n])) + (idx * (byv[n + 1] - byi[n])) - mu0 * j2[n]));
        }
        return;
    }
    double bx_km = ld_shared<NT>(bx - Bx.ks), by_km = ld_shared<NT>(by - By.ks);
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const double bx_c = ld_shared<NT>(bx), by_c = ld_shared<NT>(by), bz_c = ld_shared<NT>(bz);
        const double bz_jm = ld_shared<NT>(bz - Bz.js), bz_im = ld_shared<NT>(bz - 1);
        const double bx_jm = ld_shared<NT>(bx - Bx.js), by_im = ld_shared<NT>(by - 1);
        if (px && k >= bex.lo[2] && k < bex.hi[2])
            st_once<NT>(ex, ld_once<NT>(ex) + c2 * dt * (-(idz * (by_c - by_km)) + (idy * (bz_c - bz_jm)) - mu0 * ld_once<NT>(jx)));
        if (py && k >= bey.lo[2] && k < bey.hi[2])
            st_once<NT>(ey, ld_once<NT>(ey) + c2 * dt * (-(idx * (bz_c - bz_im)) + (idz * (bx_c - bx_km)) - mu0 * ld_once<NT>(jy)));
        if (pz && k >= bez.lo[2] && k < bez.hi[2])
            st_once<NT>(ez, ld_once<NT>(ez) + c2 * dt * (-(idy * (bx_c - bx_jm)) + (idx * (by_c - by_im)) - mu0 * ld_once<NT>(jz)));
        bx_km = bx_c; by_km = by_c;
        ex += Ex.ks; ey += Ey.ks; ez += Ez.ks; bx += Bx.ks; by += By.ks; bz += Bz.ks;
        jx += Jx.ks; jy += Jy.ks; jz += Jz.ks;
    }
}

// Production configuration: one row per workgroup, 3 planes per lane, non-temporal moves of the arrays touched once.
// MI355X, 256^3, back to back, same box (round 2; % of 8 TB/s for 72 / 96 B per cell; every one bit-identical to the CPU path):
//   2 x 4 plain 0.2290 / 0.3060 ms (65.9 / 65.8 %)   2 x 4 NT 0.2204 / 0.2866 (68.5 / 70.2)   1 x 4 NT 0.2188 / 0.2865
//   1 x 3 NT 0.2146 / 0.2803 (70.3 / 71.8)   1 x 2 NT 0.2214 / 0.2866   4 x 2 NT 0.2294 / 0.2897
//   NT on the shared operand too: 0.2633 / 0.3088 (57 / 65)   256-lane rows (TW = 4): 0.325 / 0.372 (46 / 54)
//   Round 4 (profiles/round4/r4b_stencil_loads_before_stores.txt, same box, two rounds): every load ahead of the first store
//   (BATCH) 0.2414 -> 0.2326 / 0.3081 -> 0.3031 ms (62.5 -> 64.9 / 65.3 -> 66.4 %); two planes per lane the same.
using StProduction = StencilCfg<1, 1, 3, 1, 1>;

// The first guard layer of B on every face of the brick (wxa_evolve_b_guard_layer): evolve_b_kernel's update, term by
// term in the same order (bit-identical to the valid points the neighbour brick computes), one lane per point.
struct FaceBoxes {
    int n;
    Box3 b[12];        // (face, component) boxes: one layer thick along the face's direction
    int comp[12];
    long first[13];    // running point counts
};
__global__ void __launch_bounds__(256)
evolve_b_faces_kernel(DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, FaceBoxes fb, double dt, double idx, double idy,
                      double idz) {
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= fb.first[fb.n]) return;
    int f = 0;
    while (t >= fb.first[f + 1]) ++f;
    t -= fb.first[f];
    const Box3 b = fb.b[f];
    const int n0 = b.hi[0] - b.lo[0], n1 = b.hi[1] - b.lo[1];
    const int i = b.lo[0] + (int)(t % n0), j = b.lo[1] + (int)((t / n0) % n1), k = b.lo[2] + (int)(t / ((long)n0 * n1));
    const int c = fb.comp[f];
    if (c == 0) {
        const double* ey = Ey.p + Ey.off(i, j, k);
        const double* ez = Ez.p + Ez.off(i, j, k);
        double* bx = Bx.p + Bx.off(i, j, k);
        *bx = *bx + (dt * (idz * (ey[Ey.ks] - ey[0])) - dt * (idy * (ez[Ez.js] - ez[0])));
    } else if (c == 1) {
        const double* ex = Ex.p + Ex.off(i, j, k);
        const double* ez = Ez.p + Ez.off(i, j, k);
        double* by = By.p + By.off(i, j, k);
        *by = *by + (dt * (idx * (ez[1] - ez[0])) - dt * (idz * (ex[Ex.ks] - ex[0])));
    } else {
        const double* ex = Ex.p + Ex.off(i, j, k);
        const double* ey = Ey.p + Ey.off(i, j, k);
        double* bz = Bz.p + Bz.off(i, j, k);
        *bz = *bz + (dt * (idy * (ex[Ex.js] - ex[0])) - dt * (idx * (ey[1] - ey[0])));
    }
}
#define WXA_STENCIL_DISPATCH(CALL) CALL(StProduction);

// generic box copy kernels -----------------------------------------------------
struct BoxN {
    int lo[3];
    int n[3];
};

// dst(i,j,k) = src(i + shift) for (i,j,k) in box (periodic guard fill, one direction)
__global__ void __launch_bounds__(256)
shift_copy_kernel(DevF f, BoxN box, int s0, int s1, int s2) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % box.n[0]);
        const int b = (int)((t / box.n[0]) % box.n[1]);
        const int c = (int)(t / ((long)box.n[0] * box.n[1]));
        const int i = box.lo[0] + a, j = box.lo[1] + b, k = box.lo[2] + c;
        f.p[f.off(i, j, k)] = f.p[f.off(i + s0, j + s1, k + s2)];
    }
}

// ---- PEC field boundary (Source/BoundaryConditions/WarpX_PEC.cpp) --------------------------
struct PecGeom {
    int dom_lo[3], dom_hi[3];   // cell-centred domain box, both inclusive
    int pec_lo[3], pec_hi[3];   // which faces are PEC
    int nodal[3];               // staggering of the component
};

// SetEfieldOnPEC (:117-196) for IS_E, SetBfieldOnPEC (:256-331) otherwise, on the points of `box`.
// The two rules differ only in which components are "flagged" at a face: the tangential ones for E,
// the normal one for B.  A flagged component is zero on the face (if it lives on it) and odd across
// it, the others are even.  The mirror point is strictly inside the domain along every PEC
// direction, so applying the rule slab by slab (one launch per PEC face) is idempotent on the
// points that two slabs share.
template <bool IS_E>
__global__ void __launch_bounds__(256)
apply_pec_kernel(DevF f, int icomp, BoxN box, PecGeom pg) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int ijk[3];
        ijk[0] = box.lo[0] + (int)(t % box.n[0]);
        ijk[1] = box.lo[1] + (int)((t / box.n[0]) % box.n[1]);
        ijk[2] = box.lo[2] + (int)(t / ((long)box.n[0] * box.n[1]));
        int mirror[3] = {ijk[0], ijk[1], ijk[2]};
        bool on_face = false, guard = false;
        double sign = 1.0;
#pragma unroll
        for (int idim = 0; idim < 3; ++idim) {
#pragma unroll
            for (int iside = 0; iside < 2; ++iside) {
                if (!(iside == 0 ? pg.pec_lo[idim] : pg.pec_hi[idim])) continue;
                const bool flagged = IS_E ? (icomp != idim) : (icomp == idim);
                // get_cell_count_to_boundary (:41-49)
                const int ig = iside == 0 ? pg.dom_lo[idim] - ijk[idim]
                                          : ijk[idim] - (pg.dom_hi[idim] + pg.nodal[idim]);
                if (ig == 0) {
                    if (flagged && pg.nodal[idim] == 1) on_face = true;
                } else if (ig > 0) {
                    mirror[idim] = iside == 0 ? pg.dom_lo[idim] + ig - (1 - pg.nodal[idim]) : pg.dom_hi[idim] + 1 - ig;
                    guard = true;
                    if (flagged) sign = -sign;
                }
            }
        }
        if (on_face) f(ijk[0], ijk[1], ijk[2]) = 0.0;
        else if (guard) f(ijk[0], ijk[1], ijk[2]) = sign * f(mirror[0], mirror[1], mirror[2]);
    }
}

// SetRhoOrJfieldFromPEC (:354-420) for one J component over its valid box, PEC walls with absorbing
// particle boundaries.  Every valid point touches only itself and its own mirror guard cells, so one
// thread per valid point is race-free; points farther than the guard depth from every PEC wall do
// nothing (their mirror lies outside the array), which is most of them: no memory traffic there.
__global__ void __launch_bounds__(256)
apply_pec_j_kernel(DevF f, int icomp, Box3 vb, PecGeom pg) {
    const int n0 = vb.hi[0] - vb.lo[0], n1 = vb.hi[1] - vb.lo[1], n2 = vb.hi[2] - vb.lo[2];
    const long total = (long)n0 * n1 * n2;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int ijk[3];
        ijk[0] = vb.lo[0] + (int)(t % n0);
        ijk[1] = vb.lo[1] + (int)((t / n0) % n1);
        ijk[2] = vb.lo[2] + (int)(t / ((long)n0 * n1));
        const int flo[3] = {f.lo0, f.lo1, f.lo2}, fn[3] = {f.n0, f.n1, f.n2};
        // mirror index along d for side s, or INT_MIN if that wall is not PEC / the mirror is outside the array
        int mir[3][2];
        bool any = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mir[d][s] = INT_MIN;
                if (!(s == 0 ? pg.pec_lo[d] : pg.pec_hi[d])) continue;
                // the domain box made nodal: [dom_lo, dom_hi + 1]  (:729-733, :800-805)
                const int fac = (s == 0 ? 2 * pg.dom_lo[d] : 2 * (pg.dom_hi[d] + 1)) - (1 - pg.nodal[d]);
                const int m = fac - ijk[d];
                if (m == ijk[d] || (m >= flo[d] && m < flo[d] + fn[d])) { mir[d][s] = m; any = true; }
            }
        }
        if (!any) continue;
        double v = f(ijk[0], ijk[1], ijk[2]);
        // 1) fold the guard deposits onto the interior point (zero on the wall)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int m = mir[d][s];
                if (m == INT_MIN) continue;
                if (m == ijk[d]) { v = 0.0; continue; }
                int q[3] = {ijk[0], ijk[1], ijk[2]};
                q[d] = m;
                v += (icomp != d ? -1.0 : 1.0) * f(q[0], q[1], q[2]);
            }
        }
        f(ijk[0], ijk[1], ijk[2]) = v;
        // 2) the guard cells take the image of the updated interior value
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int m = mir[d][s];
                if (m == INT_MIN || m == ijk[d]) continue;
                int q[3] = {ijk[0], ijk[1], ijk[2]};
                q[d] = m;
                f(q[0], q[1], q[2]) = icomp != d ? -v : v;
            }
        }
    }
}

// ---- moving window: WarpX::shiftMF (Source/Utils/WarpXMovingWindow.cpp:478-648) ------------------------
__global__ void __launch_bounds__(256)
set_box_kernel(DevF f, BoxN box, double value) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = box.lo[0] + (int)(t % box.n[0]);
        const int j = box.lo[1] + (int)((t / box.n[0]) % box.n[1]);
        const int k = box.lo[2] + (int)(t / ((long)box.n[0] * box.n[1]));
        f.p[f.off(i, j, k)] = value;
    }
}

// dst(i,j,k) = src(i + s0, j + s1, k + s2) on box; src and dst are different arrays of the same shape
__global__ void __launch_bounds__(256)
shifted_copy_kernel(DevF dst, DevF src, BoxN box, int s0, int s1, int s2) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = box.lo[0] + (int)(t % box.n[0]);
        const int j = box.lo[1] + (int)((t / box.n[0]) % box.n[1]);
        const int k = box.lo[2] + (int)(t / ((long)box.n[0] * box.n[1]));
        dst.p[dst.off(i, j, k)] = src.p[src.off(i + s0, j + s1, k + s2)];
    }
}

// Both guard slabs of one direction in one launch: box_lo takes src(i + shift), box_hi takes
// src(i - shift).  The boxes have the same extents; sources are valid points, destinations guard
// points, so the two halves are independent.
__global__ void __launch_bounds__(256)
shift_copy_sides_kernel(DevF f, BoxN box_lo, BoxN box_hi, int s0, int s1, int s2) {
    const long half = (long)box_lo.n[0] * box_lo.n[1] * box_lo.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * half; t += (long)gridDim.x * blockDim.x) {
        const bool hi = t >= half;
        const long u = hi ? t - half : t;
        const BoxN& box = hi ? box_hi : box_lo;
        const int sg = hi ? -1 : 1;
        const int a = (int)(u % box.n[0]);
        const int b = (int)((u / box.n[0]) % box.n[1]);
        const int c = (int)(u / ((long)box.n[0] * box.n[1]));
        const int i = box.lo[0] + a, j = box.lo[1] + b, k = box.lo[2] + c;
        f.p[f.off(i, j, k)] = f.p[f.off(i + sg * s0, j + sg * s1, k + sg * s2)];
    }
}

// SumBoundary along direction d: every residue class mod nc is summed over its members in
// [s0,s1) and the total written to all members in the allocation.  Only the classes with more
// than one member inside the allocation can change (the n[d]-nc = stag+2*ng lowest points and
// their images); the interior of the array is never touched.
__global__ void __launch_bounds__(256)
sum_periodic_kernel(DevF f, int d, int nc, int s0, int s1) {
    const int lo[3] = {f.lo0, f.lo1, f.lo2};
    const int n[3] = {f.n0, f.n1, f.n2};
    const long st[3] = {1, f.js, f.ks};
    const int da = d == 0 ? 1 : 0;           // faster of the other two
    const int db = d == 2 ? 1 : 2;           // slower of the other two
    const int nr = n[d] - nc;                // classes with >= 2 members in the allocation
    if (nr <= 0) return;
    // thread space: (fast a, class r, slow b); for d == 0 the class index is made the fastest so
    // that neighbouring lanes touch neighbouring addresses
    const long total = (long)n[da] * nr * n[db];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int a, r, b;
        if (d == 0) {
            r = (int)(t % nr); a = (int)((t / nr) % n[da]); b = (int)(t / ((long)nr * n[da]));
        } else {
            a = (int)(t % n[da]); r = (int)((t / n[da]) % nr); b = (int)(t / ((long)n[da] * nr));
        }
        double* base = f.p + a * st[da] + b * st[db];
        const int a0 = lo[d], a1 = lo[d] + n[d];
        const int first = a0 + r;
        double sum = 0.0;
        for (int m = first; m < a1; m += nc)
            if (m >= s0 && m < s1) sum += base[(long)(m - a0) * st[d]];
        for (int m = first; m < a1; m += nc) base[(long)(m - a0) * st[d]] = sum;
    }
}

// The two kernels above for several fields at once (blockIdx.y = field): the periodic fill of E and B before the gather is
// 18 launches of a few microseconds as one launch per field and direction, and every launch costs about as much again in
// the gap behind it; with the six components of a direction in one launch it is 3.  Same arithmetic, same order per field.
constexpr int WXA_MULTI_MAX = 6;
struct SideCopySet {
    DevF f[WXA_MULTI_MAX];
    BoxN lo[WXA_MULTI_MAX], hi[WXA_MULTI_MAX];
    int shift[WXA_MULTI_MAX];   // the period along d in points
};
__global__ void __launch_bounds__(256)
shift_copy_sides_multi_kernel(SideCopySet a, int d) {
    const int c = blockIdx.y;
    const DevF f = a.f[c];
    const BoxN box_lo = a.lo[c], box_hi = a.hi[c];
    const int s0 = d == 0 ? a.shift[c] : 0, s1 = d == 1 ? a.shift[c] : 0, s2 = d == 2 ? a.shift[c] : 0;
    const long half = (long)box_lo.n[0] * box_lo.n[1] * box_lo.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * half; t += (long)gridDim.x * blockDim.x) {
        const bool hi = t >= half;
        const long u = hi ? t - half : t;
        const BoxN& box = hi ? box_hi : box_lo;
        const int sg = hi ? -1 : 1;
        const int aa = (int)(u % box.n[0]);
        const int bb = (int)((u / box.n[0]) % box.n[1]);
        const int cc = (int)(u / ((long)box.n[0] * box.n[1]));
        const int i = box.lo[0] + aa, j = box.lo[1] + bb, k = box.lo[2] + cc;
        f.p[f.off(i, j, k)] = f.p[f.off(i + sg * s0, j + sg * s1, k + sg * s2)];
    }
}
struct PeriodicSumSet {
    DevF f[WXA_MULTI_MAX];
    int nc[WXA_MULTI_MAX], s0[WXA_MULTI_MAX], s1[WXA_MULTI_MAX];
};
__global__ void __launch_bounds__(256)
sum_periodic_multi_kernel(PeriodicSumSet a, int d) {
    const int c = blockIdx.y;
    const DevF f = a.f[c];
    const int nc = a.nc[c], s0 = a.s0[c], s1 = a.s1[c];
    const int lo[3] = {f.lo0, f.lo1, f.lo2};
    const int n[3] = {f.n0, f.n1, f.n2};
    const long st[3] = {1, f.js, f.ks};
    const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
    const int nr = n[d] - nc;
    if (nr <= 0) return;
    const long total = (long)n[da] * nr * n[db];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        int aa, r, b;
        if (d == 0) {
            r = (int)(t % nr); aa = (int)((t / nr) % n[da]); b = (int)(t / ((long)nr * n[da]));
        } else {
            aa = (int)(t % n[da]); r = (int)((t / n[da]) % nr); b = (int)(t / ((long)n[da] * nr));
        }
        double* base = f.p + aa * st[da] + b * st[db];
        const int a0 = lo[d], a1 = lo[d] + n[d];
        const int first = a0 + r;
        double sum = 0.0;
        for (int m = first; m < a1; m += nc)
            if (m >= s0 && m < s1) sum += base[(long)(m - a0) * st[d]];
        for (int m = first; m < a1; m += nc) base[(long)(m - a0) * st[d]] = sum;
    }
}

// T = double: the slab as it is; T = float: warpx.do_single_precision_comms (ablastr/utils/Communication.cpp:37-56,
// 90-106,159-170) -- the wire carries comm_float_type, the arrays stay double
template <class T>
__global__ void __launch_bounds__(256)
pack_kernel(DevF f, BoxN box, T* __restrict__ buf) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % box.n[0]);
        const int b = (int)((t / box.n[0]) % box.n[1]);
        const int c = (int)(t / ((long)box.n[0] * box.n[1]));
        buf[t] = static_cast<T>(f.p[f.off(box.lo[0] + a, box.lo[1] + b, box.lo[2] + c)]);
    }
}

template <class T>
__global__ void __launch_bounds__(256)
unpack_kernel(DevF f, BoxN box, const T* __restrict__ buf, int mode) {
    const long total = (long)box.n[0] * box.n[1] * box.n[2];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % box.n[0]);
        const int b = (int)((t / box.n[0]) % box.n[1]);
        const int c = (int)(t / ((long)box.n[0] * box.n[1]));
        double* q = &f.p[f.off(box.lo[0] + a, box.lo[1] + b, box.lo[2] + c)];
        if (mode == 0) *q = static_cast<double>(buf[t]); else *q += static_cast<double>(buf[t]);
    }
}

// ---- algo.maxwell_solver = ckc ---------------------------------------------------------------------------------
// CartesianCKCAlgorithm::UpwardDx / UpwardDy / UpwardDz (CartesianCKCAlgorithm.H:129-160,183-214,237-272), term by
// term in the reference's order (this file is compiled without FMA contraction: bit-identical to the CPU path).
struct CkcCoefs { double x[5], y[5], z[5]; };

template <class Acc>
__device__ __forceinline__ double ckc_up_x(const Acc& F, const double* c, int i, int j, int k) {
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
template <class Acc>
__device__ __forceinline__ double ckc_up_y(const Acc& F, const double* c, int i, int j, int k) {
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
template <class Acc>
__device__ __forceinline__ double ckc_up_z(const Acc& F, const double* c, int i, int j, int k) {
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

__device__ inline bool in_box(const Box3& b, int i, int j, int k) {
    return i >= b.lo[0] && i < b.hi[0] && j >= b.lo[1] && j < b.hi[1] && k >= b.lo[2] && k < b.hi[2];
}

// EvolveBCartesian<CartesianCKCAlgorithm> (EvolveB.cpp:164-186).  (A plain version -- one lane per point, the 3 x 18
// neighbour reads of a point through L1 / L2 -- took 1.04 ms at 256^3; it is in the history of this file.)
// The update with the E planes staged in LDS.  A workgroup owns 64 x TJ points of KC consecutive
// planes; it keeps four planes of each E component, with one halo point on every side in i and j, in a ring: while
// plane k is computed from k-1, k, k+1, plane k+2 replaces plane k-2 (one barrier per plane).  PIPE: the values of
// plane k+2 are loaded into registers before plane k is computed and written to LDS after it, so the loads' latency
// hides behind the arithmetic.  Every E value then comes from HBM / L2 once per tile (+ halo) instead of up to 16 times
// from L1; B is read and written once with non-temporal accesses.  The sums are the ones of ckc_up_x/y/z on an
// accessor into the ring: bit-identical to the plain kernel and to the CPU path.
// MI355X, 256^3, back to back (72 B/cell; Yee's EvolveB 0.216 ms = 69.8 % of 8 TB/s on the same box):
//   plain 1.038 ms (14.5 %)   TJ x KC = 8 x 16 0.363 (41.6 %)   8 x 16 PIPE 0.337 (44.8 %)   8 x 32 PIPE 0.370
//   4 x 32 PIPE 0.343   16 x 16 PIPE 0.373   8 x 8 PIPE 0.335 (45.1 %)
// (Also tried, on the Yee kernels too: the three old values of B -- E and J in evolve_e_kernel -- loaded before the first
// store instead of load - add - store per component: EvolveE 0.316 -> 0.351 ms, this kernel 0.337 -> 0.415.)
// Tried and rejected: the staging loop as `for (a = tid; a < PLANE; a += NT) slot[a] = load` (one load in flight per lane:
// 0.553); a 3 x 3 x 3 register window per component shifted along k, fed from global memory (0.689) or from a two-slot
// LDS copy of the plane (0.749: the shifts); all six derivatives evaluated before the three stores so that the LDS
// reads common to two of them are issued once (0.648: registers).  What is left is LDS read rate: 108 ds_read_b64 per point.
template <int TJ_, int KC_, int PIPE_>
struct CkcCfg {
    static constexpr int TI = 64, TJ = TJ_, KC = KC_, PIPE = PIPE_;
    static constexpr int NI = TI + 2, NJ = TJ + 2, PLANE = NI * NJ, NT = TI * TJ;
    static constexpr int PER = (PLANE + NT - 1) / NT;   // staged points per lane and plane
};

template <class CFG>
struct CkcRing {
    const double* s;   // 4 slots of CFG::PLANE
    int i0, j0;        // global index of the local point (1, 1)
    __device__ __forceinline__ double operator()(int i, int j, int k) const {
        return s[(k & 3) * CFG::PLANE + (j - j0 + 1) * CFG::NI + (i - i0 + 1)];
    }
};

// plane kk of F around the tile, halo included, into registers / from registers into a slot; points outside the
// array read 0 (they feed no valid output: a valid point's neighbours lie within the guard point the entry requires)
template <class CFG>
__device__ __forceinline__ void ckc_fetch_plane(double (&r)[CFG::PER], const DevF& F, int i0, int j0, int kk, int tid) {
    const bool kin = kk >= F.lo2 && kk < F.lo2 + F.n2;
#pragma unroll
    for (int n = 0; n < CFG::PER; ++n) {
        const int a = tid + n * CFG::NT;
        const int i = i0 - 1 + a % CFG::NI, j = j0 - 1 + a / CFG::NI;
        const bool in = a < CFG::PLANE && kin && i >= F.lo0 && i < F.lo0 + F.n0 && j >= F.lo1 && j < F.lo1 + F.n1;
        r[n] = in ? F.p[F.off(i, j, kk)] : 0.0;
    }
}
template <class CFG>
__device__ __forceinline__ void ckc_put_plane(double* slot, const double (&r)[CFG::PER], int tid) {
#pragma unroll
    for (int n = 0; n < CFG::PER; ++n) {
        const int a = tid + n * CFG::NT;
        if (a < CFG::PLANE) slot[a] = r[n];
    }
}

template <class CFG>
__global__ void __launch_bounds__(CFG::NT)
evolve_b_ckc_tiled_kernel(DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, Box3 ub, Box3 bbx, Box3 bby, Box3 bbz,
                          TileGrid tg, double dt, CkcCoefs c) {
    constexpr int PLANE = CFG::PLANE;
    __shared__ double ring[3][4 * PLANE];
    const long tile = xcd_tile_id(blockIdx.x, tg.ntiles);
    if (tile >= tg.ntiles) return;
    const int ti = (int)(tile % tg.nti);
    const int tj = (int)((tile / tg.nti) % tg.ntj);
    const int tk = (int)(tile / ((long)tg.nti * tg.ntj));
    const int i0 = ub.lo[0] + ti * CFG::TI, j0 = ub.lo[1] + tj * CFG::TJ;
    const int k0 = ub.lo[2] + tk * CFG::KC;
    const int k1 = min(k0 + CFG::KC, ub.hi[2]);
    const int tid = (int)(threadIdx.y * CFG::TI + threadIdx.x);
    const int i = i0 + (int)threadIdx.x, j = j0 + (int)threadIdx.y;
    double rx[CFG::PER], ry[CFG::PER], rz[CFG::PER];
    for (int k = k0 - 1; k <= k0 + 1; ++k) {
        ckc_fetch_plane<CFG>(rx, Ex, i0, j0, k, tid);
        ckc_fetch_plane<CFG>(ry, Ey, i0, j0, k, tid);
        ckc_fetch_plane<CFG>(rz, Ez, i0, j0, k, tid);
        ckc_put_plane<CFG>(ring[0] + (k & 3) * PLANE, rx, tid);
        ckc_put_plane<CFG>(ring[1] + (k & 3) * PLANE, ry, tid);
        ckc_put_plane<CFG>(ring[2] + (k & 3) * PLANE, rz, tid);
    }
    const CkcRing<CFG> ex{ring[0], i0, j0}, ey{ring[1], i0, j0}, ez{ring[2], i0, j0};
    const bool px = in_ij(bbx, i, j), py = in_ij(bby, i, j), pz = in_ij(bbz, i, j);
    // PIPE == 2: plane k+3 is requested while plane k is computed and plane k+2 (requested one step earlier, held in a
    // second set of registers) goes to LDS after it: two plane steps for a load to arrive instead of one -- with one
    // barrier per plane and ~1000 cycles of arithmetic per plane a single step is shorter than the HBM latency under load
    double qx[CFG::PER], qy[CFG::PER], qz[CFG::PER];
    if constexpr (CFG::PIPE == 2) {
        if (k0 + 2 <= k1) {
            ckc_fetch_plane<CFG>(rx, Ex, i0, j0, k0 + 2, tid);
            ckc_fetch_plane<CFG>(ry, Ey, i0, j0, k0 + 2, tid);
            ckc_fetch_plane<CFG>(rz, Ez, i0, j0, k0 + 2, tid);
        }
    }
    for (int k = k0; k < k1; ++k) {
        __syncthreads();   // planes k-1, k, k+1 are in the ring; everyone is done with plane k-2
        const bool more = k + 2 <= k1;   // plane k+2 for the next step, into the slot of k-2
        if constexpr (CFG::PIPE == 2) {
            if (k + 3 <= k1) {
                ckc_fetch_plane<CFG>(qx, Ex, i0, j0, k + 3, tid);
                ckc_fetch_plane<CFG>(qy, Ey, i0, j0, k + 3, tid);
                ckc_fetch_plane<CFG>(qz, Ez, i0, j0, k + 3, tid);
            }
        } else if (more) {
            ckc_fetch_plane<CFG>(rx, Ex, i0, j0, k + 2, tid);
            ckc_fetch_plane<CFG>(ry, Ey, i0, j0, k + 2, tid);
            ckc_fetch_plane<CFG>(rz, Ez, i0, j0, k + 2, tid);
            if constexpr (!CFG::PIPE) {
                ckc_put_plane<CFG>(ring[0] + ((k + 2) & 3) * PLANE, rx, tid);
                ckc_put_plane<CFG>(ring[1] + ((k + 2) & 3) * PLANE, ry, tid);
                ckc_put_plane<CFG>(ring[2] + ((k + 2) & 3) * PLANE, rz, tid);
            }
        }
        if (px && k >= bbx.lo[2] && k < bbx.hi[2]) {
            double* b = &Bx(i, j, k);
            __builtin_nontemporal_store(__builtin_nontemporal_load(b) + (dt * ckc_up_z(ey, c.z, i, j, k) - dt * ckc_up_y(ez, c.y, i, j, k)), b);
        }
        if (py && k >= bby.lo[2] && k < bby.hi[2]) {
            double* b = &By(i, j, k);
            __builtin_nontemporal_store(__builtin_nontemporal_load(b) + (dt * ckc_up_x(ez, c.x, i, j, k) - dt * ckc_up_z(ex, c.z, i, j, k)), b);
        }
        if (pz && k >= bbz.lo[2] && k < bbz.hi[2]) {
            double* b = &Bz(i, j, k);
            __builtin_nontemporal_store(__builtin_nontemporal_load(b) + (dt * ckc_up_y(ex, c.y, i, j, k) - dt * ckc_up_x(ey, c.x, i, j, k)), b);
        }
        if constexpr (CFG::PIPE) {
            if (more) {
                ckc_put_plane<CFG>(ring[0] + ((k + 2) & 3) * PLANE, rx, tid);
                ckc_put_plane<CFG>(ring[1] + ((k + 2) & 3) * PLANE, ry, tid);
                ckc_put_plane<CFG>(ring[2] + ((k + 2) & 3) * PLANE, rz, tid);
            }
        }
        if constexpr (CFG::PIPE == 2) {
#pragma unroll
            for (int n = 0; n < CFG::PER; ++n) { rx[n] = qx[n]; ry[n] = qy[n]; rz[n] = qz[n]; }
        }
    }
}

// Round 4 (profiles/round4/r4i_ckc_planes_two_steps_ahead.txt, 256^3, two rounds): 8 x 16 PIPE 1 (production until then)
// 0.3379 ms (44.7 %); the same tile with the planes requested two steps ahead 0.3378 -- the loads' latency is not what
// limits it; 4 rows x 16 planes, two steps ahead (38 KB of LDS: four workgroups per CU) 0.3172 ms (47.6 %), 4 x 32 0.341.
using CkcProduction = CkcCfg<4, 16, 2>;
// Source/Filter/Filter.cpp:105-133 with the 1-pass stencil of BilinearFilter.cpp:26-60 ({0.25, 0.25} per direction);
// same tap order as the reference -> bit-identical.  The staging of the CKC kernel above: a workgroup filters
// 64 x TJ points of KC planes from a ring of four staged input planes (halo of one point, zero padding beyond the
// allocated box, Filter.cpp:109-114), the next plane's loads in flight while the current one is filtered, one barrier
// per plane; every input point comes from HBM / L2 ~1.3 times instead of 27 times through L1.
// (Round 1's version -- three planes, `for (a = tid; ...) lds = load`, two barriers per plane -- took 0.130 ms per
// component at 256^3 + guards.)
using FilterCfg = CkcCfg<8, 16, 1>;

// Filter::DoFilter (Source/Filter/Filter.cpp:92-133) for any stencil lengths: the filter that the NCI corrector applies to
// E and B before the gather (NCIGodfreyFilter: lengths 1, 1, 5, along z only; PhysicalParticleContainer::applyNCIFilter,
// PhysicalParticleContainer.cpp:2097-2172).  One lane per point over the whole allocation, zero padding beyond it, the
// reference's loop and term order (sss = s0 s1 s2; eight mirrored taps summed left to right) -> bit-identical to the
// CPU path.  Six fields per species and step in the boosted-frame runs only: not tiled.
struct FilterStencils {
    static constexpr int MAXLEN = 8;
    double s[3][MAXLEN];
    int n[3];
};
__global__ void __launch_bounds__(256)
filter_stencil_kernel(DevF src, DevF dst, FilterStencils fs) {
    const long total = (long)src.n0 * src.n1 * src.n2;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = src.lo0 + (int)(t % src.n0);
        const int j = src.lo1 + (int)((t / src.n0) % src.n1);
        const int k = src.lo2 + (int)(t / ((long)src.n0 * src.n1));
        auto zp = [&](int ii, int jj, int kk) -> double {
            return (ii >= src.lo0 && ii < src.lo0 + src.n0 && jj >= src.lo1 && jj < src.lo1 + src.n1 && kk >= src.lo2 &&
                    kk < src.lo2 + src.n2) ? src.p[src.off(ii, jj, kk)] : 0.0;
        };
        double d = 0.0;
        for (int i2 = 0; i2 < fs.n[2]; ++i2)
            for (int i1 = 0; i1 < fs.n[1]; ++i1)
                for (int i0 = 0; i0 < fs.n[0]; ++i0) {
                    const double sss = fs.s[0][i0] * fs.s[1][i1] * fs.s[2][i2];
                    d += sss * (zp(i - i0, j - i1, k - i2) + zp(i + i0, j - i1, k - i2) + zp(i - i0, j + i1, k - i2) +
                                zp(i + i0, j + i1, k - i2) + zp(i - i0, j - i1, k + i2) + zp(i + i0, j - i1, k + i2) +
                                zp(i - i0, j + i1, k + i2) + zp(i + i0, j + i1, k + i2));
                }
        dst.p[dst.off(i, j, k)] = d;
    }
}

// The same for stencil lengths (1, 1, N2) -- the NCI corrector's Godfrey filter, six fields per species and step of a
// boosted-frame run: a lane marches KC points up its column with the 2 N2 - 1 values it needs in registers, so that every
// point is read once (+ 2 (N2 - 1) / KC at the segment ends) instead of 2 N2 - 1 times through the caches (0.90 ms per field at
// 256 x 256 x 512, 7 % of the HBM rate: 5.4 of BASELINE config 5's 94 ms per step, profiles/round5/README.md).  The
// arithmetic is the generic kernel's, term by term: with i0 = i1 = 0 the eight mirrored taps are the point below four
// times and the point above four times, summed left to right.
template <int N2, int KC>
__global__ void __launch_bounds__(256)
filter_stencil_z_kernel(DevF src, DevF dst, FilterStencils fs) {
    const long columns = (long)src.n0 * src.n1;
    const int nseg = (src.n2 + KC - 1) / KC;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= columns * nseg) return;
    const int i = src.lo0 + (int)(t % src.n0);
    const int j = src.lo1 + (int)((t / src.n0) % src.n1);
    const int k0 = src.lo2 + (int)(t / columns) * KC, k1 = min(k0 + KC, src.lo2 + src.n2);
    auto zp = [&](int kk) -> double { return (kk >= src.lo2 && kk < src.lo2 + src.n2) ? src.p[src.off(i, j, kk)] : 0.0; };
    double w[2 * N2 - 1];   // w[m] = the value at k - (N2 - 1) + m
#pragma unroll
    for (int m = 0; m < 2 * N2 - 1; ++m) w[m] = zp(k0 - (N2 - 1) + m);
    double sss[N2];
#pragma unroll
    for (int i2 = 0; i2 < N2; ++i2) sss[i2] = fs.s[0][0] * fs.s[1][0] * fs.s[2][i2];
    for (int k = k0; k < k1; ++k) {
        const double next = zp(k + N2);
        double d = 0.0;
#pragma unroll
        for (int i2 = 0; i2 < N2; ++i2) {
            const double a = w[N2 - 1 - i2], b = w[N2 - 1 + i2];
            d += sss[i2] * (a + a + a + a + b + b + b + b);
        }
        dst.p[dst.off(i, j, k)] = d;
#pragma unroll
        for (int m = 0; m + 1 < 2 * N2 - 1; ++m) w[m] = w[m + 1];
        w[2 * N2 - 2] = next;
    }
}

template <class CFG>
__global__ void __launch_bounds__(CFG::NT)
filter_bilinear_kernel(DevF src, DevF dst, TileGrid tg) {
    constexpr int PLANE = CFG::PLANE;
    __shared__ double ring[4 * PLANE];
    const long tile = xcd_tile_id(blockIdx.x, tg.ntiles);
    if (tile >= tg.ntiles) return;
    const int ti = (int)(tile % tg.nti), tj = (int)((tile / tg.nti) % tg.ntj), tk = (int)(tile / ((long)tg.nti * tg.ntj));
    const int i0 = src.lo0 + ti * CFG::TI, j0 = src.lo1 + tj * CFG::TJ, k0 = src.lo2 + tk * CFG::KC;
    const int hi0 = src.lo0 + src.n0, hi1 = src.lo1 + src.n1, hi2 = src.lo2 + src.n2;
    const int k1 = min(k0 + CFG::KC, hi2);
    const int tid = (int)(threadIdx.y * CFG::TI + threadIdx.x);
    const int i = i0 + (int)threadIdx.x, j = j0 + (int)threadIdx.y;
    double r[CFG::PER];
    for (int k = k0 - 1; k <= k0 + 1; ++k) {
        ckc_fetch_plane<CFG>(r, src, i0, j0, k, tid);
        ckc_put_plane<CFG>(ring + (k & 3) * PLANE, r, tid);
    }
    const CkcRing<CFG> in{ring, i0, j0};
    const bool active = i < hi0 && j < hi1;
    for (int k = k0; k < k1; ++k) {
        __syncthreads();   // planes k-1, k, k+1 are in the ring; everyone is done with plane k-2
        const bool more = k + 2 <= k1;
        if (more) ckc_fetch_plane<CFG>(r, src, i0, j0, k + 2, tid);
        if (active) {
            auto tap = [&](int di, int dj, int dk) -> double { return in(i + di, j + dj, k + dk); };
            double d = 0.0;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
                    for (int i0_ = 0; i0_ < 2; ++i0_) {
                        const double sss = 0.25 * 0.25 * 0.25;
                        d += sss * (tap(-i0_, -i1, -i2) + tap(+i0_, -i1, -i2) + tap(-i0_, +i1, -i2) +
                                    tap(+i0_, +i1, -i2) + tap(-i0_, -i1, +i2) + tap(+i0_, -i1, +i2) +
                                    tap(-i0_, +i1, +i2) + tap(+i0_, +i1, +i2));
                    }
            dst.p[dst.off(i, j, k)] = d;
        }
        if (more) ckc_put_plane<CFG>(ring + ((k + 2) & 3) * PLANE, r, tid);
    }
}

static inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static bool box_inside(const wxa_field_view& v, const int32_t blo[3], const int32_t bhi[3]) {
    for (int d = 0; d < 3; ++d)
        if (blo[d] < v.lo[d] || bhi[d] > v.lo[d] + v.n[d] || bhi[d] < blo[d]) return false;
    return true;
}

}  // namespace wxa

using namespace wxa;

extern "C" {

wxa_status wxa_evolve_b(const wxa_field_view E[3], const wxa_field_view B[3], double dt,
                        const double dinv[3], void* stream) {
    WXA_REQUIRE(E && B && dinv, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(E[c]) && view_ok(B[c]), "bad field view");
    if (!yee_E(E) || !yee_B(B)) {
        set_last_error("wxa_evolve_b: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) WXA_REQUIRE(E[c].ng[d] >= 1, "EvolveB needs >= 1 guard point on E");
    const Box3 bx = valid_box(B[0]), by = valid_box(B[1]), bz = valid_box(B[2]);
    Box3 ub;
    for (int d = 0; d < 3; ++d) {
        ub.lo[d] = std::min(bx.lo[d], std::min(by.lo[d], bz.lo[d]));
        ub.hi[d] = std::max(bx.hi[d], std::max(by.hi[d], bz.hi[d]));
    }
#define WXA_LAUNCH_B(CFG)                                                                                       \
    do {                                                                                                        \
        const TileGrid tg = make_tiles_cfg<CFG>(ub);                                                            \
        if (tg.ntiles > 0)                                                                                      \
            hipLaunchKernelGGL((evolve_b_kernel<CFG>), dim3((unsigned)xcd_grid_size(tg.ntiles)),                \
                               dim3(CFG::TI, CFG::TJ), 0, (hipStream_t)stream, make_devf(E[0]), make_devf(E[1]), \
                               make_devf(E[2]), make_devf(B[0]), make_devf(B[1]), make_devf(B[2]), ub, bx, by,  \
                               bz, tg, dt, dinv[0], dinv[1], dinv[2]);                                          \
    } while (0)
    WXA_STENCIL_DISPATCH(WXA_LAUNCH_B)
#undef WXA_LAUNCH_B
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// The first guard layer of B next to the faces of the directions with grow[d] != 0, updated like the valid
// points (EvolveB.cpp:164-186) from the guard points of E and B already present: the points EvolveE reads
// beyond the valid box are those of the components cell-centred along d, at the indices lo - 1 and
// lo + ncell (faces only).  Replaces the FillBoundaryB that follows the update (wxa_evolve_b_guard_layer).
wxa_status wxa_evolve_b_guard_layer(const wxa_field_view E[3], const wxa_field_view B[3], double dt,
                                    const double dinv[3], const int32_t grow[3], void* stream) {
    WXA_REQUIRE(E && B && dinv && grow, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(E[c]) && view_ok(B[c]), "bad field view");
    if (!yee_E(E) || !yee_B(B)) {
        set_last_error("wxa_evolve_b_guard_layer: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d)
            WXA_REQUIRE(!grow[d] || (E[c].ng[d] >= 2 && B[c].ng[d] >= 1), "guard layer update needs 2 guard points on E, 1 on B");
    // One launch for all faces (round 6; until then one launch of the tiled kernel per face: six thin boxes, the two x
    // faces on 1 lane in 64): a lane per guard point of a component, the boxes listed in the kernel argument.  The layer
    // of a face spans the component's valid range along the other two directions, so no two boxes share a point.
    FaceBoxes fb;
    fb.n = 0;
    fb.first[0] = 0;
    for (int d = 0; d < 3; ++d) {
        if (!grow[d]) continue;
        for (int side = 0; side < 2; ++side)
            for (int c = 0; c < 3; ++c) {
                if (B[c].stag[d]) continue;   // nodal along d: no guard point is read
                Box3 b = valid_box(B[c]);
                const int layer = side == 0 ? b.lo[d] - 1 : b.hi[d];
                b.lo[d] = layer; b.hi[d] = layer + 1;
                const long pts = (long)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
                if (pts <= 0) continue;
                fb.b[fb.n] = b; fb.comp[fb.n] = c;
                fb.first[fb.n + 1] = fb.first[fb.n] + pts;
                ++fb.n;
            }
    }
    if (fb.n > 0 && fb.first[fb.n] > 0)
        hipLaunchKernelGGL(evolve_b_faces_kernel, dim3((unsigned)((fb.first[fb.n] + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           make_devf(E[0]), make_devf(E[1]), make_devf(E[2]), make_devf(B[0]), make_devf(B[1]), make_devf(B[2]),
                           fb, dt, dinv[0], dinv[1], dinv[2]);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// CartesianCKCAlgorithm::InitializeStencilCoefficients, 3-D branch (CartesianCKCAlgorithm.H:36-101)
void wxa_ckc_stencil_coefficients(const double cell_size[3], double cx[5], double cy[5], double cz[5]) {
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
double wxa_ckc_max_dt(const double cell_size[3]) {
    return std::min(cell_size[0], std::min(cell_size[1], cell_size[2])) / PhysConst::c;
}

wxa_status wxa_evolve_b_ckc(const wxa_field_view E[3], const wxa_field_view B[3], double dt, const double cx[5],
                            const double cy[5], const double cz[5], void* stream) {
    WXA_REQUIRE(E && B && cx && cy && cz, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(E[c]) && view_ok(B[c]), "bad field view");
    if (!yee_E(E) || !yee_B(B)) {
        set_last_error("wxa_evolve_b_ckc: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) WXA_REQUIRE(E[c].ng[d] >= 1, "the CKC update of B needs >= 1 guard point on E");
    const Box3 bx = valid_box(B[0]), by = valid_box(B[1]), bz = valid_box(B[2]);
    Box3 ub;
    for (int d = 0; d < 3; ++d) {
        ub.lo[d] = std::min(bx.lo[d], std::min(by.lo[d], bz.lo[d]));
        ub.hi[d] = std::max(bx.hi[d], std::max(by.hi[d], bz.hi[d]));
        if (ub.hi[d] <= ub.lo[d]) return WXA_OK;
    }
    CkcCoefs cc;
    for (int n = 0; n < 5; ++n) { cc.x[n] = cx[n]; cc.y[n] = cy[n]; cc.z[n] = cz[n]; }
    {
#define WXA_CKC_LAUNCH(CFG)                                                                                            \
    do {                                                                                                               \
        const TileGrid tg = make_tiles_cfg<StencilCfg<1, CFG::TJ, CFG::KC, 1>>(ub);                                    \
        hipLaunchKernelGGL(evolve_b_ckc_tiled_kernel<CFG>, dim3((unsigned)xcd_grid_size(tg.ntiles)), dim3(CFG::TI, CFG::TJ), \
                           0, (hipStream_t)stream, make_devf(E[0]), make_devf(E[1]), make_devf(E[2]), make_devf(B[0]),    \
                           make_devf(B[1]), make_devf(B[2]), ub, bx, by, bz, tg, dt, cc);                              \
    } while (0)
        WXA_CKC_LAUNCH(CkcProduction);
#undef WXA_CKC_LAUNCH
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_evolve_e(const wxa_field_view E[3], const wxa_field_view B[3], const wxa_field_view J[3],
                        double dt, const double dinv[3], void* stream) {
    WXA_REQUIRE(E && B && J && dinv, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(E[c]) && view_ok(B[c]) && view_ok(J[c]), "bad field view");
    if (!yee_E(E) || !yee_B(B) || !yee_E(J)) {
        set_last_error("wxa_evolve_e: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) WXA_REQUIRE(B[c].ng[d] >= 1, "EvolveE needs >= 1 guard point on B");
    const Box3 bx = valid_box(E[0]), by = valid_box(E[1]), bz = valid_box(E[2]);
    for (int c = 0; c < 3; ++c) {
        const Box3 bj = valid_box(J[c]);
        const Box3 be = valid_box(E[c]);
        for (int d = 0; d < 3; ++d)
            WXA_REQUIRE(bj.lo[d] == be.lo[d] && bj.hi[d] == be.hi[d], "J and E valid boxes differ");
    }
    Box3 ub;
    for (int d = 0; d < 3; ++d) {
        ub.lo[d] = std::min(bx.lo[d], std::min(by.lo[d], bz.lo[d]));
        ub.hi[d] = std::max(bx.hi[d], std::max(by.hi[d], bz.hi[d]));
    }
#define WXA_LAUNCH_E(CFG)                                                                                       \
    do {                                                                                                        \
        const TileGrid tg = make_tiles_cfg<CFG>(ub);                                                            \
        if (tg.ntiles > 0)                                                                                      \
            hipLaunchKernelGGL((evolve_e_kernel<CFG>), dim3((unsigned)xcd_grid_size(tg.ntiles)),                \
                               dim3(CFG::TI, CFG::TJ), 0, (hipStream_t)stream, make_devf(E[0]), make_devf(E[1]), \
                               make_devf(E[2]), make_devf(B[0]), make_devf(B[1]), make_devf(B[2]),              \
                               make_devf(J[0]), make_devf(J[1]), make_devf(J[2]), ub, bx, by, bz, tg, dt,       \
                               dinv[0], dinv[1], dinv[2]);                                                      \
    } while (0)
    WXA_STENCIL_DISPATCH(WXA_LAUNCH_E)
#undef WXA_LAUNCH_E
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_filter_bilinear(const wxa_field_view* src, const wxa_field_view* dst, void* stream) {
    WXA_REQUIRE(src && dst && view_ok(*src) && view_ok(*dst), "bad field view");
    WXA_REQUIRE(src->p != dst->p, "src and dst must not alias");
    for (int d = 0; d < 3; ++d)
        WXA_REQUIRE(src->lo[d] == dst->lo[d] && src->n[d] == dst->n[d], "src/dst boxes differ");
    TileGrid tg;
    tg.nti = (src->n[0] + FilterCfg::TI - 1) / FilterCfg::TI;
    tg.ntj = (src->n[1] + FilterCfg::TJ - 1) / FilterCfg::TJ;
    tg.ntk = (src->n[2] + FilterCfg::KC - 1) / FilterCfg::KC;
    tg.ntiles = (long)tg.nti * tg.ntj * tg.ntk;
    hipLaunchKernelGGL(filter_bilinear_kernel<FilterCfg>, dim3((unsigned)xcd_grid_size(tg.ntiles)),
                       dim3(FilterCfg::TI, FilterCfg::TJ), 0, (hipStream_t)stream, make_devf(*src), make_devf(*dst), tg);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_filter_stencil(const wxa_field_view* src, const wxa_field_view* dst, const double* s0, int32_t n0,
                              const double* s1, int32_t n1, const double* s2, int32_t n2, void* stream) {
    WXA_REQUIRE(src && dst && view_ok(*src) && view_ok(*dst) && s0 && s1 && s2, "bad argument");
    WXA_REQUIRE(src->p != dst->p, "src and dst must not alias");
    WXA_REQUIRE(n0 >= 1 && n0 <= FilterStencils::MAXLEN && n1 >= 1 && n1 <= FilterStencils::MAXLEN && n2 >= 1 &&
                n2 <= FilterStencils::MAXLEN, "stencil length");
    for (int d = 0; d < 3; ++d)
        WXA_REQUIRE(src->lo[d] == dst->lo[d] && src->n[d] == dst->n[d], "src/dst boxes differ");
    FilterStencils fs;
    fs.n[0] = n0; fs.n[1] = n1; fs.n[2] = n2;
    for (int a = 0; a < FilterStencils::MAXLEN; ++a) {
        fs.s[0][a] = a < n0 ? s0[a] : 0.0; fs.s[1][a] = a < n1 ? s1[a] : 0.0; fs.s[2][a] = a < n2 ? s2[a] : 0.0;
    }
    const long total = (long)src->n[0] * src->n[1] * src->n[2];
    if (total == 0) return WXA_OK;
    if (n0 == 1 && n1 == 1 && n2 == 5) {   // the Godfrey stencil
        constexpr int KC = 32;
        const long lanes = (long)src->n[0] * src->n[1] * ((src->n[2] + KC - 1) / KC);
        hipLaunchKernelGGL((filter_stencil_z_kernel<5, KC>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, make_devf(*src), make_devf(*dst), fs);
        WXA_LAUNCH_CHECK();
        return WXA_OK;
    }
    hipLaunchKernelGGL(filter_stencil_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, make_devf(*src),
                       make_devf(*dst), fs);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_fill_boundary_periodic(const wxa_field_view* f, const int ng[3], const int periodic[3],
                                      void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && ng && periodic, "bad argument");
    const DevF df = make_devf(*f);
    int lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
        WXA_REQUIRE(ng[d] <= f->ng[d], "ng exceeds allocated guards");
        lo[d] = f->lo[d] + f->ng[d];
        hi[d] = f->lo[d] + f->n[d] - f->ng[d];
    }
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d] || ng[d] <= 0) continue;
        const int nc = f->n[d] - 2 * f->ng[d] - f->stag[d];
        WXA_REQUIRE(ng[d] <= nc, "guard depth exceeds the period");
        const int v0 = f->lo[d] + f->ng[d], v1 = f->lo[d] + f->n[d] - f->ng[d];
        BoxN blo, bhi;
        for (int e = 0; e < 3; ++e) { blo.lo[e] = bhi.lo[e] = lo[e]; blo.n[e] = bhi.n[e] = hi[e] - lo[e]; }
        blo.lo[d] = v0 - ng[d]; bhi.lo[d] = v1;
        blo.n[d] = bhi.n[d] = ng[d];
        int sh[3] = {0, 0, 0};
        sh[d] = nc;
        const long total = 2 * (long)blo.n[0] * blo.n[1] * blo.n[2];
        if (total > 0)
            hipLaunchKernelGGL(shift_copy_sides_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, df,
                               blo, bhi, sh[0], sh[1], sh[2]);
        lo[d] = v0 - ng[d];
        hi[d] = v1 + ng[d];
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_fill_boundary_periodic_multi(const wxa_field_view* f, int32_t nf, const int ng[3], const int periodic[3],
                                            void* stream) {
    WXA_REQUIRE(f && ng && periodic && nf >= 1 && nf <= WXA_MULTI_MAX, "bad argument");
    int lo[WXA_MULTI_MAX][3], hi[WXA_MULTI_MAX][3];
    for (int c = 0; c < nf; ++c) {
        WXA_REQUIRE(view_ok(f[c]), "bad field view");
        for (int d = 0; d < 3; ++d) {
            WXA_REQUIRE(ng[d] <= f[c].ng[d], "ng exceeds allocated guards");
            lo[c][d] = f[c].lo[d] + f[c].ng[d];
            hi[c][d] = f[c].lo[d] + f[c].n[d] - f[c].ng[d];
        }
    }
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d] || ng[d] <= 0) continue;
        SideCopySet a;
        long most = 0;
        for (int c = 0; c < nf; ++c) {
            const int nc = f[c].n[d] - 2 * f[c].ng[d] - f[c].stag[d];
            WXA_REQUIRE(ng[d] <= nc, "guard depth exceeds the period");
            const int v0 = f[c].lo[d] + f[c].ng[d], v1 = f[c].lo[d] + f[c].n[d] - f[c].ng[d];
            a.f[c] = make_devf(f[c]);
            for (int e = 0; e < 3; ++e) { a.lo[c].lo[e] = a.hi[c].lo[e] = lo[c][e]; a.lo[c].n[e] = a.hi[c].n[e] = hi[c][e] - lo[c][e]; }
            a.lo[c].lo[d] = v0 - ng[d]; a.hi[c].lo[d] = v1;
            a.lo[c].n[d] = a.hi[c].n[d] = ng[d];
            a.shift[c] = nc;
            most = std::max(most, 2 * (long)a.lo[c].n[0] * a.lo[c].n[1] * a.lo[c].n[2]);
            lo[c][d] = v0 - ng[d];
            hi[c][d] = v1 + ng[d];
        }
        for (int c = nf; c < WXA_MULTI_MAX; ++c) { a.f[c] = a.f[0]; a.lo[c] = a.lo[0]; a.hi[c] = a.hi[0]; a.shift[c] = 0; }
        if (most > 0)
            hipLaunchKernelGGL(shift_copy_sides_multi_kernel, dim3(grid_for(most), (unsigned)nf), dim3(256), 0,
                               (hipStream_t)stream, a, d);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_sum_boundary_periodic_multi(const wxa_field_view* f, int32_t nf, const int src_ng[3], const int periodic[3],
                                           void* stream) {
    WXA_REQUIRE(f && src_ng && periodic && nf >= 1 && nf <= WXA_MULTI_MAX, "bad argument");
    for (int c = 0; c < nf; ++c) WXA_REQUIRE(view_ok(f[c]), "bad field view");
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d]) continue;
        PeriodicSumSet a;
        long most = 0;
        const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
        for (int c = 0; c < nf; ++c) {
            WXA_REQUIRE(src_ng[d] <= f[c].ng[d], "src_ng exceeds allocated guards");
            const int nc = f[c].n[d] - 2 * f[c].ng[d] - f[c].stag[d];
            WXA_REQUIRE(nc >= f[c].stag[d] + 2 * f[c].ng[d], "brick thinner than its guard cells");
            a.f[c] = make_devf(f[c]);
            a.nc[c] = nc;
            a.s0[c] = f[c].lo[d] + f[c].ng[d] - src_ng[d];
            a.s1[c] = f[c].lo[d] + f[c].n[d] - f[c].ng[d] + src_ng[d];
            most = std::max(most, (long)f[c].n[da] * (f[c].n[d] - nc) * f[c].n[db]);
        }
        for (int c = nf; c < WXA_MULTI_MAX; ++c) { a.f[c] = a.f[0]; a.nc[c] = a.nc[0]; a.s0[c] = a.s0[0]; a.s1[c] = a.s1[0]; }
        hipLaunchKernelGGL(sum_periodic_multi_kernel, dim3(grid_for(most), (unsigned)nf), dim3(256), 0, (hipStream_t)stream,
                           a, d);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_sync_nodal_periodic(const wxa_field_view* f, const int periodic[3], void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && periodic, "bad argument");
    const DevF df = make_devf(*f);
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d] || !f->stag[d]) continue;
        const int nc = f->n[d] - 2 * f->ng[d] - f->stag[d];
        BoxN b;
        for (int e = 0; e < 3; ++e) { b.lo[e] = f->lo[e] + f->ng[e]; b.n[e] = f->n[e] - 2 * f->ng[e]; }
        b.lo[d] = f->lo[d] + f->ng[d] + nc;
        b.n[d] = 1;
        int s[3] = {0, 0, 0};
        s[d] = -nc;
        const long total = (long)b.n[0] * b.n[1] * b.n[2];
        if (total <= 0) continue;
        hipLaunchKernelGGL(shift_copy_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, df, b,
                           s[0], s[1], s[2]);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_sum_boundary_periodic(const wxa_field_view* f, const int src_ng[3], const int periodic[3],
                                     void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && src_ng && periodic, "bad argument");
    const DevF df = make_devf(*f);
    for (int d = 0; d < 3; ++d) {
        if (!periodic[d]) continue;
        WXA_REQUIRE(src_ng[d] <= f->ng[d], "src_ng exceeds allocated guards");
        const int nc = f->n[d] - 2 * f->ng[d] - f->stag[d];
        const int s0 = f->lo[d] + f->ng[d] - src_ng[d];
        const int s1 = f->lo[d] + f->n[d] - f->ng[d] + src_ng[d];
        WXA_REQUIRE(nc >= f->stag[d] + 2 * f->ng[d], "brick thinner than its guard cells");
        const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
        const long total = (long)f->n[da] * (f->n[d] - nc) * f->n[db];
        hipLaunchKernelGGL(sum_periodic_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, df, d,
                           nc, s0, s1);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

}  // extern "C"

template <class T>
static wxa_status pack_box_t(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], T* buf, void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && buf, "bad argument");
    WXA_REQUIRE(box_inside(*f, blo, bhi), "box outside the allocation");
    BoxN b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = blo[d]; b.n[d] = bhi[d] - blo[d]; }
    const long total = (long)b.n[0] * b.n[1] * b.n[2];
    if (total == 0) return WXA_OK;
    hipLaunchKernelGGL(pack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, make_devf(*f), b,
                       buf);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

template <class T>
static wxa_status unpack_box_t(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], const T* buf, int mode,
                               void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && buf, "bad argument");
    WXA_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (copy) or 1 (add)");
    WXA_REQUIRE(box_inside(*f, blo, bhi), "box outside the allocation");
    BoxN b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = blo[d]; b.n[d] = bhi[d] - blo[d]; }
    const long total = (long)b.n[0] * b.n[1] * b.n[2];
    if (total == 0) return WXA_OK;
    hipLaunchKernelGGL(unpack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, make_devf(*f), b,
                       buf, mode);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

extern "C" {

wxa_status wxa_pack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], double* buf,
                        void* stream) { return pack_box_t<double>(f, blo, bhi, buf, stream); }
wxa_status wxa_unpack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                          const double* buf, int mode, void* stream) { return unpack_box_t<double>(f, blo, bhi, buf, mode, stream); }
wxa_status wxa_pack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3], float* buf,
                            void* stream) { return pack_box_t<float>(f, blo, bhi, buf, stream); }
wxa_status wxa_unpack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                              const float* buf, int mode, void* stream) { return unpack_box_t<float>(f, blo, bhi, buf, mode, stream); }

}  // extern "C"

// ApplyPECtoEfield (:457-538) / ApplyPECtoBfield (:540-626): the rule above on tilebox(ixType, ng);
// only the slabs at and behind the PEC faces can change, so only they are launched.
template <bool IS_E>
static wxa_status apply_pec(const wxa_field_view F[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                            const int32_t pec_lo[3], const int32_t pec_hi[3], const int32_t ng[3], void* stream) {
    WXA_REQUIRE(F && dom_lo && dom_hi && pec_lo && pec_hi && ng, "null argument");
    for (int c = 0; c < 3; ++c) {
        WXA_REQUIRE(view_ok(F[c]), "bad field view");
        for (int d = 0; d < 3; ++d) {
            WXA_REQUIRE(ng[d] >= 0 && ng[d] <= F[c].ng[d], "ng exceeds allocated guards");
            WXA_REQUIRE(dom_hi[d] >= dom_lo[d], "empty domain");
        }
    }
    for (int c = 0; c < 3; ++c) {
        const wxa_field_view& f = F[c];
        PecGeom pg;
        int glo[3], ghi[3];   // tilebox(ixType, ng) of this brick, hi exclusive
        for (int d = 0; d < 3; ++d) {
            pg.dom_lo[d] = dom_lo[d]; pg.dom_hi[d] = dom_hi[d];
            pg.pec_lo[d] = pec_lo[d] ? 1 : 0; pg.pec_hi[d] = pec_hi[d] ? 1 : 0;
            pg.nodal[d] = f.stag[d];
            glo[d] = f.lo[d] + f.ng[d] - ng[d];
            ghi[d] = f.lo[d] + f.n[d] - f.ng[d] + ng[d];
        }
        const DevF df = make_devf(f);
        for (int d = 0; d < 3; ++d)
            for (int side = 0; side < 2; ++side) {
                if (!(side == 0 ? pg.pec_lo[d] : pg.pec_hi[d])) continue;
                BoxN b;
                for (int e = 0; e < 3; ++e) { b.lo[e] = glo[e]; b.n[e] = ghi[e] - glo[e]; }
                // points with ig >= 0: at or beyond the face
                const int lo_d = side == 0 ? glo[d] : std::max(glo[d], dom_hi[d] + f.stag[d]);
                const int hi_d = side == 0 ? std::min(ghi[d], dom_lo[d] + 1) : ghi[d];
                b.lo[d] = lo_d; b.n[d] = hi_d - lo_d;
                const long total = (long)b.n[0] * b.n[1] * b.n[2];
                if (b.n[d] <= 0 || total <= 0) continue;   // this brick does not touch the face
                hipLaunchKernelGGL((apply_pec_kernel<IS_E>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                   df, c, b, pg);
            }
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_shift_field_window(const wxa_field_view* f, double* tmp, int32_t dir, int32_t num_shift,
                                  const int periodic[3], void* stream) {
    WXA_REQUIRE(f && view_ok(*f) && tmp && periodic, "bad argument");
    WXA_REQUIRE(dir >= 0 && dir < 3, "dir must be 0..2");
    WXA_REQUIRE(num_shift >= 0 && num_shift <= f->ng[dir], "shift exceeds the guard depth");
    WXA_REQUIRE(tmp != f->p, "the scratch array must be a different array");
    if (num_shift == 0) return WXA_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t count = (size_t)f->kstride * (size_t)f->n[2];
    // MultiFab::Copy(tmpmf, mf, 0, 0, nc, ng)
    WXA_HIP_CHECK(hipMemcpyAsync(tmp, f->p, sizeof(double) * count, hipMemcpyDeviceToDevice, st));
    wxa_field_view tv = *f;
    tv.p = tmp;
    // FillBoundary(tmpmf, ng_mw, periodicity): one guard cell, num_shift along the window direction
    int ng_mw[3] = {1, 1, 1};
    ng_mw[dir] = num_shift;
    for (int d = 0; d < 3; ++d) ng_mw[d] = std::min(ng_mw[d], (int)f->ng[d]);
    const bool keep_guards = periodic[dir] == WXA_WINDOW_KEEP_GUARDS;   // the next brick's cells are in the guards already
    const int per[3] = {dir == 0 ? 0 : periodic[0], dir == 1 ? 0 : periodic[1], dir == 2 ? 0 : periodic[2]};
    wxa_status rc = wxa_fill_boundary_periodic(&tv, ng_mw, per, stream);
    if (rc != WXA_OK) return rc;
    const DevF dsrc = make_devf(tv), ddst = make_devf(*f);
    // everything beyond the domain on the high side takes the external field (0)
    BoxN z;
    for (int d = 0; d < 3; ++d) { z.lo[d] = f->lo[d]; z.n[d] = f->n[d]; }
    z.lo[dir] = f->lo[dir] + f->n[dir] - f->ng[dir];
    z.n[dir] = f->ng[dir];
    long total = (long)z.n[0] * z.n[1] * z.n[2];
    if (total > 0 && !keep_guards)
        hipLaunchKernelGGL(set_box_kernel, dim3(grid_for(total)), dim3(256), 0, st, dsrc, z, 0.0);
    // dst(i) = src(i + shift) on the array box shrunk by num_shift on the high side
    BoxN b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = f->lo[d]; b.n[d] = f->n[d]; }
    b.n[dir] -= num_shift;
    int sh[3] = {0, 0, 0};
    sh[dir] = num_shift;
    total = (long)b.n[0] * b.n[1] * b.n[2];
    if (total > 0)
        hipLaunchKernelGGL(shifted_copy_kernel, dim3(grid_for(total)), dim3(256), 0, st, ddst, dsrc, b, sh[0], sh[1], sh[2]);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// one launch over the valid box of f; icomp = the vector component f is (sign +1 across the walls it is
// normal to), or 3 for a scalar that is tangential to every wall (rho)
static wxa_status pec_reflect(const wxa_field_view& f, int icomp, const int32_t dom_lo[3], const int32_t dom_hi[3],
                              const int32_t pec_lo[3], const int32_t pec_hi[3], bool transverse_guards, void* stream) {
    PecGeom pg;
    Box3 vb;
    for (int d = 0; d < 3; ++d) {
        WXA_REQUIRE(dom_hi[d] >= dom_lo[d], "empty domain");
        pg.dom_lo[d] = dom_lo[d]; pg.dom_hi[d] = dom_hi[d];
        pg.pec_lo[d] = pec_lo[d] ? 1 : 0; pg.pec_hi[d] = pec_hi[d] ? 1 : 0;
        pg.nodal[d] = f.stag[d];
        // the guard columns of a wall-free direction are folded too when asked (rho: before the guard sum)
        const bool grow = transverse_guards && !pec_lo[d] && !pec_hi[d];
        vb.lo[d] = grow ? f.lo[d] : f.lo[d] + f.ng[d];
        vb.hi[d] = grow ? f.lo[d] + f.n[d] : f.lo[d] + f.n[d] - f.ng[d];
    }
    const long total = (long)(vb.hi[0] - vb.lo[0]) * (vb.hi[1] - vb.lo[1]) * (vb.hi[2] - vb.lo[2]);
    if (total <= 0) return WXA_OK;
    hipLaunchKernelGGL(apply_pec_j_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, make_devf(f), icomp,
                       vb, pg);
    return WXA_OK;
}

wxa_status wxa_apply_pec_j(const wxa_field_view J[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                           const int32_t pec_lo[3], const int32_t pec_hi[3], void* stream) {
    WXA_REQUIRE(J && dom_lo && dom_hi && pec_lo && pec_hi, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(J[c]), "bad field view");
    for (int c = 0; c < 3; ++c) {
        const wxa_status rc = pec_reflect(J[c], c, dom_lo, dom_hi, pec_lo, pec_hi, false, stream);
        if (rc != WXA_OK) return rc;
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_apply_pec_rho(const wxa_field_view* rho, const int32_t dom_lo[3], const int32_t dom_hi[3],
                             const int32_t pec_lo[3], const int32_t pec_hi[3], void* stream) {
    WXA_REQUIRE(rho && dom_lo && dom_hi && pec_lo && pec_hi, "null argument");
    WXA_REQUIRE(view_ok(*rho), "bad field view");
    const wxa_status rc = pec_reflect(*rho, 3, dom_lo, dom_hi, pec_lo, pec_hi, true, stream);
    if (rc != WXA_OK) return rc;
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_apply_pec_e(const wxa_field_view E[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                           const int32_t pec_lo[3], const int32_t pec_hi[3], const int32_t ng[3], void* stream) {
    return apply_pec<true>(E, dom_lo, dom_hi, pec_lo, pec_hi, ng, stream);
}

wxa_status wxa_apply_pec_b(const wxa_field_view B[3], const int32_t dom_lo[3], const int32_t dom_hi[3],
                           const int32_t pec_lo[3], const int32_t pec_hi[3], const int32_t ng[3], void* stream) {
    return apply_pec<false>(B, dom_lo, dom_hi, pec_lo, pec_hi, ng, stream);
}

// Several arrays zeroed by one launch (the three components of J at the top of every step: as three hipMemsetAsync they
// were nine fill dispatches -- head, body and tail of each -- in the step's timeline).  blockIdx.y = array; 16-byte
// non-temporal stores (the arrays are not read again before the deposition's atomics, a whole step's particle traffic later).
struct ZeroSet {
    double* p[6];
    long n[6];   // doubles
};
__global__ void __launch_bounds__(256)
zero_multi_kernel(ZeroSet z) {
    typedef double D2 __attribute__((ext_vector_type(2)));
    const int f = blockIdx.y;
    double* p = z.p[f];
    long n = z.n[f];
    if (n <= 0) return;
    if (reinterpret_cast<uintptr_t>(p) & 8u) {   // a caller's array that starts on an odd double: one scalar store in front
        if (blockIdx.x == 0 && threadIdx.x == 0) p[0] = 0.0;
        ++p; --n;
    }
    const long n2 = n >> 1;
    D2* q = reinterpret_cast<D2*>(p);
    const D2 zero = {0.0, 0.0};
    // four 16-byte stores per lane, each a whole 4 KB row of the workgroup (the first version -- one non-temporal store
    // per trip of a grid-stride loop -- reached 3.5 TB/s, half of what the runtime's fill kernel does)
    const long base = (long)blockIdx.x * (4 * 256) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long t = base + u * 256;
        if (t < n2) q[t] = zero;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}

extern "C" {

wxa_status wxa_field_set_zero(const wxa_field_view* f, void* stream) {
    WXA_REQUIRE(f && view_ok(*f), "bad field view");
    WXA_HIP_CHECK(hipMemsetAsync(f->p, 0, sizeof(double) * (size_t)f->kstride * f->n[2], (hipStream_t)stream));
    return WXA_OK;
}

wxa_status wxa_field_set_zero_multi(const wxa_field_view* f, int32_t nf, void* stream) {
    WXA_REQUIRE(f && nf >= 1 && nf <= 6, "1 to 6 fields");
    ZeroSet z{};
    long most = 0;
    for (int c = 0; c < nf; ++c) {
        WXA_REQUIRE(view_ok(f[c]), "bad field view");
        z.p[c] = f[c].p;
        z.n[c] = (long)f[c].kstride * f[c].n[2];
        most = std::max(most, z.n[c]);
    }
    if (most == 0) return WXA_OK;
    const long blocks = (most / 2 + 1023) / 1024 + 1;   // 1024 16-byte stores per workgroup (+ 1: an array that starts on an odd double)
    WXA_REQUIRE(blocks < (1L << 31), "array too long");
    hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks, (unsigned)nf), dim3(256), 0, (hipStream_t)stream, z);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

}  // extern "C"
