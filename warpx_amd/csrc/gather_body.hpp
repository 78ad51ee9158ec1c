// Gather bodies shared by the global-load kernel (particles.hip) and the LDS-tile kernel
// (gather_tile.hip).  doGatherShapeN<O,G> (Source/Particles/Gather/FieldGather.H:36-424)
// specialised to the Yee index types: per direction only two weight arrays occur, the order-O
// nodal one and the order-(O-G) cell-centred one (:98-121,135-158,171-194).
#ifndef WXA_GATHER_BODY_HPP_
#define WXA_GATHER_BODY_HPP_

#include "shapes.hpp"

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
