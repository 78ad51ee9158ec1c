// Gather bodies shared by the global-load kernel (particles.hip) and the LDS-tile kernel
// (gather_tile.hip).  doGatherShapeN<O,G> (Source/Particles/Gather/FieldGather.H:36-424)
// specialised to the Yee index types: per direction only two weight arrays occur, the order-O
// nodal one and the order-(O-G) cell-centred one (:98-121,135-158,171-194).
#ifndef WXA_GATHER_BODY_HPP_
#define WXA_GATHER_BODY_HPP_

#include "shapes.hpp"

#include <type_traits>

namespace wxa {

struct PV {
    double* __restrict__ x; double* __restrict__ y; double* __restrict__ z; double* __restrict__ w;
    double* __restrict__ ux; double* __restrict__ uy; double* __restrict__ uz;
    uint64_t* __restrict__ id;
    long np;
};
static inline PV make_pv(const wxa_particle_view& p) {
    return PV{p.x, p.y, p.z, p.w, p.ux, p.uy, p.uz, p.idcpu, (long)p.np};
}
static inline bool pv_ok(const wxa_particle_view* p) {
    return p && p->np >= 0 && (p->np == 0 || (p->x && p->y && p->z && p->w && p->ux && p->uy && p->uz));
}

template <int O, int G>
struct GatherShapes {
    double sxn[O + 1], sxc[O + 1 - G], syn[O + 1], syc[O + 1 - G], szn[O + 1], szc[O + 1 - G];
    int jn, jc, kn, kc, ln, lc;   // global grid index of the leftmost point, node / cell centring
};

template <int O, int G>
__device__ __forceinline__ void gather_shapes(double xp, double yp, double zp, const Geom& g,
                                              GatherShapes<O, G>& s) {
    const double x = (xp - g.xmin) * g.dxi;
    const double y = (yp - g.ymin) * g.dyi;
    const double z = (zp - g.zmin) * g.dzi;
    s.jn = g.lo0 + shape_factor<O>(s.sxn, x);
    s.jc = g.lo0 + shape_factor<O - G>(s.sxc, x - 0.5);
    s.kn = g.lo1 + shape_factor<O>(s.syn, y);
    s.kc = g.lo1 + shape_factor<O - G>(s.syc, y - 0.5);
    s.ln = g.lo2 + shape_factor<O>(s.szn, z);
    s.lc = g.lo2 + shape_factor<O - G>(s.szc, z - 0.5);
}

// sum_{iz,iy,ix} sx[ix] sy[iy] sz[iz] F(ix,iy,iz) over an NX x NY x NZ block starting at `base`
template <int NX, int NY, int NZ>
__device__ __forceinline__ double gather_rows(const double* __restrict__ base, long js, long ks,
                                              const double* __restrict__ sx, const double* __restrict__ sy,
                                              const double* __restrict__ sz) {
    double acc = 0.0;
#pragma unroll
    for (int iz = 0; iz < NZ; ++iz) {
        double plane = 0.0;
#pragma unroll
        for (int iy = 0; iy < NY; ++iy) {
            const double* __restrict__ row = base + iy * js + iz * ks;
            double r = 0.0;
#pragma unroll
            for (int ix = 0; ix < NX; ++ix) r += sx[ix] * row[ix];
            // sz (sum_y sy r): one fma per row and one per plane.  (sy sz) r is what the reference writes; its products
            // sy sz are common to several components and the compiler kept all 49 of them alive (125 VGPRs)
            plane += sy[iy] * r;
        }
        acc += sz[iz] * plane;
    }
    return acc;
}

// The same sum from an LDS tile (row stride JS, plane stride KS, compile time) with single ds_read_b64 reads.
// hipcc pairs the reads of a row into ds_read2_b64, which the LDS serves at half the rate of ds_read_b64 (8 cycles per
// wave instruction for two points against 2 per point, MI355X_MICROARCH.md) and whose 8-bit offsets reach only 2 KB, so
// that every other read needs an address add of its own (123 v_add_u32 per particle at order 3).  Nothing in C++ keeps
// the reads single AND lets the compiler schedule them: volatile or atomic reads are issued two at a time with a full
// wait each (no reordering of ordered accesses), plain reads are paired.  So the reads are inline asm on a fixed
// software pipeline -- the rows of the block in order, RB rows in flight ahead of the row being summed -- with one
// s_waitcnt per row whose count is the number of this function's reads issued after that row (LDS operations return in
// order; anything else on the counter only makes the wait longer).  The "+v" operands of the wait make every use of a
// row's values depend on it.
// tests/hipcpu (host compiler, no LDS) defines its own: plain loads, no waits.
#ifndef WXA_LDS_READ_B64
typedef unsigned wxa_lds_addr;
#define WXA_LDS_ADDR(p) ((unsigned)(size_t)(const __attribute__((address_space(3))) double*)(p))
#define WXA_LDS_READ_B64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off))
#define WXA_LDS_WAIT1(n, a) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(n))
#define WXA_LDS_WAIT2(n, a, b) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(n))
#define WXA_LDS_WAIT3(n, a, b, c) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(n))
#define WXA_LDS_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n))
#endif
template <int OFF>
__device__ __forceinline__ double lds_read_b64(const wxa_lds_addr addr) {
    double v;
    WXA_LDS_READ_B64(v, addr, OFF);
    return v;
}
template <int NX, int PENDING>
__device__ __forceinline__ void lds_wait_row(double (&v)[NX]) {
    static_assert(NX >= 1 && NX <= 4 && PENDING >= 0 && PENDING <= 15, "lgkmcnt is a 4-bit counter");
    if constexpr (NX == 1) WXA_LDS_WAIT1(PENDING, v[0]);
    else if constexpr (NX == 2) WXA_LDS_WAIT2(PENDING, v[0], v[1]);
    else if constexpr (NX == 3) WXA_LDS_WAIT3(PENDING, v[0], v[1], v[2]);
    else WXA_LDS_WAIT4(PENDING, v[0], v[1], v[2], v[3]);
}
template <int NX, int NY, int NZ, int JS, int KS, int RB>
__device__ __forceinline__ double gather_rows_lds(const double* base, const double* __restrict__ sx,
                                                  const double* __restrict__ sy, const double* __restrict__ sz) {
    constexpr int ROWS = NY * NZ;
    static_assert(RB >= 1 && RB * NX <= 15, "rows in flight");
    const wxa_lds_addr addr = WXA_LDS_ADDR(base);
    double v[ROWS][NX];   // fully unrolled: a row lives from its reads to its fma chain
    auto issue = [&](auto row_c) {
        constexpr int row = decltype(row_c)::value;
        constexpr int off = ((row % NY) * JS + (row / NY) * KS) * 8;
        v[row][0] = lds_read_b64<off>(addr);
        if constexpr (NX > 1) v[row][1] = lds_read_b64<off + 8>(addr);
        if constexpr (NX > 2) v[row][2] = lds_read_b64<off + 16>(addr);
        if constexpr (NX > 3) v[row][3] = lds_read_b64<off + 24>(addr);
    };
    double acc = 0.0, plane = 0.0;
    auto for_rows = [&](auto&& self, auto row_c) {
        constexpr int row = decltype(row_c)::value;
        if constexpr (row < ROWS) {
            if constexpr (row + RB < ROWS) issue(std::integral_constant<int, row + RB>{});
            constexpr int ahead = (ROWS - 1 - row < RB ? ROWS - 1 - row : RB) * NX;
            lds_wait_row<NX, ahead>(v[row]);
            constexpr int iy = row % NY, iz = row / NY;
            double r = 0.0;
#pragma unroll
            for (int ix = 0; ix < NX; ++ix) r += sx[ix] * v[row][ix];
            plane += sy[iy] * r;
            if constexpr (iy == NY - 1) { acc += sz[iz] * plane; plane = 0.0; }
            self(self, std::integral_constant<int, row + 1>{});
        }
    };
    // prologue: the first RB rows
    auto prologue = [&](auto&& self, auto row_c) {
        constexpr int row = decltype(row_c)::value;
        if constexpr (row < RB && row < ROWS) {
            issue(row_c);
            self(self, std::integral_constant<int, row + 1>{});
        }
    };
    prologue(prologue, std::integral_constant<int, 0>{});
    for_rows(for_rows, std::integral_constant<int, 0>{});
    return acc;
}

// Yee: Ex(c,n,n) Ey(n,c,n) Ez(n,n,c) Bx(n,c,c) By(c,n,c) Bz(c,c,n); component order of the
// reference loops (FieldGather.H:368-423): Ex, Ey, Ez, Bz, By, Bx
template <int O, int G>
__device__ __forceinline__ void gather_global(const GatherShapes<O, G>& s, const DevF& Ex, const DevF& Ey,
                                              const DevF& Ez, const DevF& Bx, const DevF& By, const DevF& Bz,
                                              double& Exp, double& Eyp, double& Ezp, double& Bxp, double& Byp,
                                              double& Bzp) {
    constexpr int NN = O + 1, NC = O + 1 - G;
    Exp = gather_rows<NC, NN, NN>(Ex.p + Ex.off(s.jc, s.kn, s.ln), Ex.js, Ex.ks, s.sxc, s.syn, s.szn);
    Eyp = gather_rows<NN, NC, NN>(Ey.p + Ey.off(s.jn, s.kc, s.ln), Ey.js, Ey.ks, s.sxn, s.syc, s.szn);
    Ezp = gather_rows<NN, NN, NC>(Ez.p + Ez.off(s.jn, s.kn, s.lc), Ez.js, Ez.ks, s.sxn, s.syn, s.szc);
    Bzp = gather_rows<NC, NC, NN>(Bz.p + Bz.off(s.jc, s.kc, s.ln), Bz.js, Bz.ks, s.sxc, s.syc, s.szn);
    Byp = gather_rows<NC, NN, NC>(By.p + By.off(s.jc, s.kn, s.lc), By.js, By.ks, s.sxc, s.syn, s.szc);
    Bxp = gather_rows<NN, NC, NC>(Bx.p + Bx.off(s.jn, s.kc, s.lc), Bx.js, Bx.ks, s.sxn, s.syc, s.szc);
}

}  // namespace wxa
#endif
