// Particle kernels for gfx950: gather + push (PushPX / PushP), current and charge
// deposition (global-atomics variants; the LDS-tile variants live in deposit_tile.hip),
// periodic wrap, counting sort by cell.
#include "deposit_body.hpp"
#include <complex>

#include "gather_body.hpp"
#include "push_sort.hpp"
#include "workspace.hpp"

#include <hipcub/hipcub.hpp>

namespace wxa {

template <int O, int G, int PUSHER, bool MOVE>
__global__ void __launch_bounds__(256)
gather_push_kernel(PV p, DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, Geom g, double q,
                   double m, double dt, ExtEB ext, PushSort hook) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    double xp = p.x[ip], yp = p.y[ip], zp = p.z[ip];
    GatherShapes<O, G> s;
    gather_shapes<O, G>(xp, yp, zp, g, s);
    double Exp, Eyp, Ezp, Bxp, Byp, Bzp;
    gather_global<O, G>(s, Ex, Ey, Ez, Bx, By, Bz, Exp, Eyp, Ezp, Bxp, Byp, Bzp);

    add_external_fields(ext, ip, Exp, Eyp, Ezp, Bxp, Byp, Bzp);
    double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip];
    push_momentum<PUSHER>(ux, uy, uz, Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, m, dt);
    if constexpr (MOVE) {
        update_position(xp, yp, zp, ux, uy, uz, dt);
        if (!push_sort_tail(hook, p, ip, xp, yp, zp, ux, uy, uz)) return;   // written to the sorted tile instead
        p.x[ip] = xp; p.y[ip] = yp; p.z[ip] = zp;
    }
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz;
}

// The repeated plasma lens seen by every particle before the push (add_lens_fields, shapes.hpp):
// out[c * stride + ip], c = Ex Ey Bx By
__global__ void __launch_bounds__(256) lens_fields_kernel(PV p, ExtLens L, double* __restrict__ out, long stride) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    double Ex = 0., Ey = 0., Bx = 0., By = 0.;
    add_lens_fields(L, p.x[ip], p.y[ip], p.z[ip], p.ux[ip], p.uy[ip], p.uz[ip], Ex, Ey, Bx, By);
    out[ip] = Ex; out[stride + ip] = Ey; out[2 * stride + ip] = Bx; out[3 * stride + ip] = By;
}

// ---------------------------------------------------------------------------
// One lane per particle, global fp64 atomics (the reference's GPU strategy,
// Source/Particles/Deposition/CurrentDeposition.H:309-334,683-824); used when no cell sort
// is available.  Bodies in deposit_body.hpp.
template <int O>
__global__ void __launch_bounds__(256)
deposit_esirkepov_global_kernel(PV p, DevF Jx, DevF Jy, DevF Jz, Geom g, double q, EsirkepovStep es) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    const ParticleState ps{p.x[ip], p.y[ip], p.z[ip], p.w[ip], p.ux[ip], p.uy[ip], p.uz[ip]};
    EsirkepovShapes<O> s;
    esirkepov_shapes<O>(ps, g, q, es, s);
    GlobalSink sink = make_global_sink(Jx, Jy, Jz);
    sink.bi = s.bi; sink.bj = s.bj; sink.bk = s.bk;
    esirkepov_accumulate<O>(s, es, sink);
}

template <int O>
__global__ void __launch_bounds__(256)
deposit_direct_global_kernel(PV p, DevF Jx, DevF Jy, DevF Jz, Geom g, double q, double relative_time) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    const ParticleState ps{p.x[ip], p.y[ip], p.z[ip], p.w[ip], p.ux[ip], p.uy[ip], p.uz[ip]};
    DirectShapes<O> s;
    direct_shapes<O>(ps, g, q, relative_time, s);
    GlobalSink sink = make_global_sink(Jx, Jy, Jz);
    direct_accumulate<O>(s, sink);
}

// Source/Particles/Deposition/ChargeDeposition.H:37-180 (3-D), rho of any staggering
template <int O>
__global__ void __launch_bounds__(256)
deposit_charge_kernel(PV p, DevF rho, int s0, int s1, int s2, Geom g, double q) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    const double invvol = g.dxi * g.dyi * g.dzi;
    const double wq = q * p.w[ip] * invvol;
    const double x = (p.x[ip] - g.xmin) * g.dxi;
    const double y = (p.y[ip] - g.ymin) * g.dyi;
    const double z = (p.z[ip] - g.zmin) * g.dzi;
    double sx[O + 1], sy[O + 1], sz[O + 1];
    const int i = g.lo0 + shape_factor<O>(sx, s0 ? x : x - 0.5);
    const int j = g.lo1 + shape_factor<O>(sy, s1 ? y : y - 0.5);
    const int k = g.lo2 + shape_factor<O>(sz, s2 ? z : z - 0.5);
    double* __restrict__ r = rho.p + rho.off(i, j, k);
#pragma unroll
    for (int iz = 0; iz <= O; iz++)
#pragma unroll
        for (int iy = 0; iy <= O; iy++)
#pragma unroll
            for (int ix = 0; ix <= O; ix++)
                atomic_add_f64(r + ix + iy * rho.js + iz * rho.ks, sx[ix] * sy[iy] * sz[iz] * wq);
}

__device__ __forceinline__ double wrap_periodic(double v, double plo, double phi) {
    const double L = phi - plo;
    if (v >= phi) {
        v -= L;
        if (v < plo) v = plo;
    } else if (v < plo) {
        v += L;
        if (v >= phi) v = nextafter(phi, plo);
    }
    return v;
}

struct PeriodicBox {
    double plo[3], phi[3];
    int on[3];
};

__global__ void __launch_bounds__(256)
enforce_periodic_kernel(double* __restrict__ x, double* __restrict__ y, double* __restrict__ z, long np,
                        PeriodicBox pb) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= np) return;
    // the three loads first: a conditional store between them would keep one load in flight at a time
    const double vx = pb.on[0] ? x[ip] : 0.0, vy = pb.on[1] ? y[ip] : 0.0, vz = pb.on[2] ? z[ip] : 0.0;
    if (pb.on[0]) { const double w = wrap_periodic(vx, pb.plo[0], pb.phi[0]); if (w != vx) x[ip] = w; }
    if (pb.on[1]) { const double w = wrap_periodic(vy, pb.plo[1], pb.phi[1]); if (w != vy) y[ip] = w; }
    if (pb.on[2]) { const double w = wrap_periodic(vz, pb.plo[2], pb.phi[2]); if (w != vz) z[ip] = w; }
}

// ---- counting sort by cell (the key: SortGeom, cell_of in push_sort.hpp) ---------------
// Histogram + rank.  The input is usually almost sorted (a few % of the particles changed cell
// since the last sort), so equal keys sit in neighbouring lanes: each run of equal keys inside a
// wave issues ONE atomic for the whole run instead of one per particle.
__global__ void __launch_bounds__(256)
sort_count_kernel(const double* __restrict__ x, const double* __restrict__ y,
                  const double* __restrict__ z, const uint64_t* __restrict__ id, long np, SortGeom s,
                  int* __restrict__ cell, int* __restrict__ rank, int* __restrict__ hist) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = ip < np;
    const uint64_t pid = (valid && id) ? id[ip] : 0;   // in flight together with the position
    int c = valid ? cell_of(s, x[ip], y[ip], z[ip]) : -1;
    if (valid && id && pid == WXA_IDCPU_RETIRED) c = s.retired_bin;
    const int prev = __shfl_up(c, 1);
    const bool head = (lane == 0) || (c != prev);
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = (2ULL << lane) - 1ULL;          // bits 0..lane (all ones at lane 63)
    const int hl = 63 - __clzll(heads & upto);                      // head lane of my run
    const unsigned long long above = heads & ~upto;
    const int nh = above ? (__ffsll((long long)above) - 1) : 64;    // first lane of the next run
    int base = 0;
    if (head && c >= 0) base = atomicAdd(&hist[c], nh - lane);
    base = __shfl(base, hl);
    if (valid) {
        if (cell) cell[ip] = c;   // null: the scatter works the key out again from the position it loads anyway
        rank[ip] = base + (lane - hl);
    }
}

// The scatter, for an input that is nearly sorted already (every sort after the first: a particle moves less than
// a cell between sorts, so its destination index is within a few hundred of its source index).  A workgroup takes
// SW_CHUNK consecutive source particles and a window of destination indices around them; a component at a time, the
// particles whose destination lies inside the window are placed at their destination offset in LDS and the window is
// written out in index order -- whole lines instead of 8-byte writes scattered over them -- and the others (movers
// beyond the margin; everything, on an unsorted input) are written directly as before.  A destination slot belongs to
// exactly one particle, so the masked window writes of neighbouring workgroups never touch the same element.
constexpr int SW_THREADS = 512;

// RECELL: the keys are not read from the array the count pass wrote (4 bytes per particle written there and read here, of
// the 152 the sort moves per particle) but worked out again from the position, which this kernel loads anyway: x, y, z and
// the id of the lane's SW_U particles are loaded first and kept for their own component passes.
template <int SW_U, int SW_MARGIN, bool RECELL = false>
__global__ void __launch_bounds__(SW_THREADS)
sort_scatter_window_kernel(PV src, PV dst, const int* __restrict__ cell, const int* __restrict__ rank,
                           const int* __restrict__ offsets, SortGeom sg = SortGeom{}) {
    constexpr int SW_CHUNK = SW_THREADS * SW_U, SW_WIN = SW_CHUNK + 2 * SW_MARGIN;
    __shared__ double win[SW_WIN];
    __shared__ unsigned char mine[SW_WIN];
    const int tid = threadIdx.x;
    const long c0 = (long)blockIdx.x * SW_CHUNK;
    const long w0 = c0 > SW_MARGIN ? c0 - SW_MARGIN : 0;
    for (int a = tid; a < SW_WIN; a += SW_THREADS) mine[a] = 0;
    long d[SW_U];
    int slot[SW_U];   // offset in the window, or -1: written directly
    int ce[SW_U], ra[SW_U];
    double keep[RECELL ? 4 : 1][SW_U];   // RECELL: x, y, z, id (as a bit pattern) of the lane's particles
#pragma unroll
    for (int u = 0; u < SW_U; ++u) {
        const long ip = c0 + u * SW_THREADS + tid;
        const bool in = ip < src.np;
        if constexpr (RECELL) {
            keep[0][u] = in ? src.x[ip] : 0.0; keep[1][u] = in ? src.y[ip] : 0.0; keep[2][u] = in ? src.z[ip] : 0.0;
            keep[3][u] = in && src.id ? reinterpret_cast<const double*>(src.id)[ip] : 0.0;
        } else {
            ce[u] = in ? cell[ip] : 0;
        }
        ra[u] = in ? rank[ip] : 0;
    }
    if constexpr (RECELL) {
#pragma unroll
        for (int u = 0; u < SW_U; ++u) {
            ce[u] = cell_of(sg, keep[0][u], keep[1][u], keep[2][u]);
            if (src.id && __builtin_bit_cast(unsigned long long, keep[3][u]) == WXA_IDCPU_RETIRED) ce[u] = sg.retired_bin;
        }
    }
#pragma unroll
    for (int u = 0; u < SW_U; ++u) {
        const long ip = c0 + u * SW_THREADS + tid;
        d[u] = ip < src.np ? (long)offsets[ce[u]] + ra[u] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SW_U; ++u) {
        const long o = d[u] - w0;
        slot[u] = (d[u] >= 0 && o >= 0 && o < SW_WIN) ? (int)o : -1;
        if (slot[u] >= 0) mine[slot[u]] = 1;
    }
    __syncthreads();
    const bool has_id = src.id && dst.id;
    const double* sp[8] = {src.x, src.y, src.z, src.w, src.ux, src.uy, src.uz, reinterpret_cast<const double*>(src.id)};
    double* dp[8] = {dst.x, dst.y, dst.z, dst.w, dst.ux, dst.uy, dst.uz, reinterpret_cast<double*>(dst.id)};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c == 7 && !has_id) break;
        double v[SW_U];
#pragma unroll
        for (int u = 0; u < SW_U; ++u) {
            const long ip = c0 + u * SW_THREADS + tid;
            if (RECELL && (c < 3 || c == 7)) v[u] = keep[c < 3 ? c : 3][u];
            else v[u] = ip < src.np ? sp[c][ip] : 0.0;   // ids travel as bit patterns
        }
#pragma unroll
        for (int u = 0; u < SW_U; ++u) {
            if (slot[u] >= 0) win[slot[u]] = v[u];
            else if (d[u] >= 0) dp[c][d[u]] = v[u];
        }
        __syncthreads();
        for (int a = tid; a < SW_WIN; a += SW_THREADS)
            if (mine[a]) dp[c][w0 + a] = win[a];
        __syncthreads();
    }
}

// ---- 3-way partition for Redistribute ---------------------------------------------
__global__ void __launch_bounds__(256)
partition_flag_kernel(const double* __restrict__ pos, long np, double lo, double hi, int* __restrict__ stay,
                      unsigned long long* __restrict__ counters) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= np) return;
    const double v = pos[ip];
    const int key = v < lo ? 1 : (v >= hi ? 2 : 0);
    stay[ip] = key == 0 ? 1 : 0;
    if (key) atomicAdd(&counters[key], 1ULL);
}

__global__ void __launch_bounds__(256)
partition_scatter_kernel(PV src, PV dst, const double* __restrict__ pos, double lo, double hi,
                         const int* __restrict__ stay_scan, long nstay, long nminus,
                         unsigned long long* __restrict__ cursors) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= src.np) return;
    const double v = pos[ip];
    long d;
    if (v < lo) d = nstay + (long)atomicAdd(&cursors[1], 1ULL);
    else if (v >= hi) d = nstay + nminus + (long)atomicAdd(&cursors[2], 1ULL);
    else d = stay_scan[ip];
    const double x = src.x[ip], y = src.y[ip], z = src.z[ip], w = src.w[ip];   // every load before the first store
    const double ux = src.ux[ip], uy = src.uy[ip], uz = src.uz[ip];
    const bool has_id = src.id && dst.id;
    const uint64_t id = has_id ? src.id[ip] : 0;
    dst.x[d] = x; dst.y[d] = y; dst.z[d] = z; dst.w[d] = w;
    dst.ux[d] = ux; dst.uy[d] = uy; dst.uz[d] = uz;
    if (has_id) dst.id[d] = id;
}

static inline unsigned blocks_for(long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

static inline wxa_particle_view tail_view(const wxa_particle_view& p, int64_t first) {
    wxa_particle_view t = p;
    t.x += first; t.y += first; t.z += first; t.w += first; t.ux += first; t.uy += first; t.uz += first;
    if (t.idcpu) t.idcpu += first;
    t.np = p.np - first;
    return t;
}

// ---- laser antenna push (LaserParticleContainer.cpp:795-951, LaserProfileGaussian.cpp:104-161) -------------
// Everything that does not depend on the particle is folded on the host into a complex prefactor
// P = e_max exp(i phase) / D * exp(-(t - t_peak)^2 / tau^2) and the complex inverse waist 1 / (w^2 D),
// D = 1 + 2 i f / (k0 w^2); the amplitude at (X, Y) is Re[P exp(-(X^2 + Y^2) / (w^2 D))].
struct LaserPushGeom {
    double position[3], p_X[3], p_Y[3];
    double mobility;
    double pre_re, pre_im;     // P
    double iw2_re, iw2_im;     // 1 / (w^2 D)
    double drift[3];           // c beta_boost nvec (0 in the lab frame)
    double gamma_boost;        // >= 1
};

__global__ void __launch_bounds__(256)
laser_push_kernel(PV p, LaserPushGeom lg, double dt) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.np) return;
    const double x = p.x[i], y = p.y[i], z = p.z[i];
    const double Xp = lg.p_X[0] * (x - lg.position[0]) + lg.p_X[1] * (y - lg.position[1]) + lg.p_X[2] * (z - lg.position[2]);
    const double Yp = lg.p_Y[0] * (x - lg.position[0]) + lg.p_Y[1] * (y - lg.position[1]) + lg.p_Y[2] * (z - lg.position[2]);
    const double r2 = Xp * Xp + Yp * Yp;
    // exp(-(r2) (a + i b)) = exp(-r2 a) (cos(r2 b) - i sin(r2 b))
    const double mag = exp(-r2 * lg.iw2_re);
    const double er = mag * cos(r2 * lg.iw2_im), ei = -mag * sin(r2 * lg.iw2_im);
    const double amplitude = lg.pre_re * er - lg.pre_im * ei;
    const double sign_charge = (p.w[i] > 0) ? -1.0 : 1.0;
    const double v_over_c = sign_charge * lg.mobility * amplitude;
    // in a boosted frame the antenna also drifts with -beta_boost c along nvec (:907-915)
    const double vx = PhysConst::c * v_over_c * lg.p_X[0] - lg.drift[0];
    const double vy = PhysConst::c * v_over_c * lg.p_X[1] - lg.drift[1];
    const double vz = PhysConst::c * v_over_c * lg.p_X[2] - lg.drift[2];
    const double gamma = lg.gamma_boost / sqrt(1.0 - v_over_c * v_over_c);
    p.ux[i] = gamma * vx; p.uy[i] = gamma * vy; p.uz[i] = gamma * vz;
    p.x[i] = x + vx * dt; p.y[i] = y + vy * dt; p.z[i] = z + vz * dt;
}

// ---- plasma injection: PhysicalParticleContainer::AddPlasma (PhysicalParticleContainer.cpp:924-1333) ---------
struct InjectGeom {
    double corner[3], dx[3], blo[3], bhi[3], lo[3], hi[3], u[3], uth[3], origin[3];
    int nc[3], ppc[3];
    double density, scale_fac;
    unsigned long long seed;
    int thermal;
    // boosted frame / ballistic correction: z0 = gamma_boost (z za - zb) (applyBallisticCorrection, :138-148),
    // za = 1 - beta_boost betaz_bulk, zb = c t (betaz_bulk - beta_boost)
    double gamma_boost, beta_boost, za, zb;
};

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so a particle's draws depend on its position only
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ double uniform53(unsigned hi, unsigned lo) {   // (0, 1)
    return ((double)((((unsigned long long)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// three standard normal draws for the injection point with the global lattice coordinates (sx, sy, sz)
__device__ __forceinline__ void normal3(unsigned long long seed, int sx, int sy, int sz, double n[3]) {
    unsigned a[4] = {(unsigned)sx, (unsigned)sy, (unsigned)sz, 0u}, b[4] = {(unsigned)sx, (unsigned)sy, (unsigned)sz, 1u};
    philox4x32_10(a, (unsigned)seed, (unsigned)(seed >> 32));
    philox4x32_10(b, (unsigned)seed, (unsigned)(seed >> 32));
    const double r0 = sqrt(-2.0 * log(uniform53(a[0], a[1]))), t0 = 2.0 * M_PI * uniform53(a[2], a[3]);
    const double r1 = sqrt(-2.0 * log(uniform53(b[0], b[1]))), t1 = 2.0 * M_PI * uniform53(b[2], b[3]);
    n[0] = r0 * cos(t0); n[1] = r0 * sin(t0); n[2] = r1 * cos(t1);
}

// one thread per lattice point; accepted points take consecutive slots (one atomic per wave)
__global__ void __launch_bounds__(256)
add_plasma_kernel(PV dst, InjectGeom ig, long npoints, unsigned long long* __restrict__ count) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = t < npoints;
    double pos[3] = {0.0, 0.0, 0.0};
    if (ok) {
        // the reference's roundings, one per operation: a fused corner + (i + r) * dx differs in the last bit, which
        // moves particles off the reference's lattice and can flip the bounds tests below at a box edge
#pragma clang fp contract(off)
        const int nppc = ig.ppc[0] * ig.ppc[1] * ig.ppc[2];
        const long cell = t / nppc;
        const int ip = (int)(t % nppc);
        const int iv[3] = {(int)(cell % ig.nc[0]), (int)((cell / ig.nc[0]) % ig.nc[1]), (int)(cell / ((long)ig.nc[0] * ig.nc[1]))};
        // InjectorPositionRegular::getPositionUnitBox (Source/Initialization/InjectorPosition.H:74-92)
        const int ny = ig.ppc[1], nz = ig.ppc[2];
        const int ix_part = ip / (ny * nz);
        const int iz_part = (ip - ix_part * (ny * nz)) / ny;
        const int iy_part = (ip - ix_part * (ny * nz)) - ny * iz_part;
        const double r[3] = {(0.5 + ix_part) / ig.ppc[0], (0.5 + iy_part) / ig.ppc[1], (0.5 + iz_part) / ig.ppc[2]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double clo = ig.corner[d] + (iv[d] + 0.0) * ig.dx[d], chi = ig.corner[d] + (iv[d] + 1.0) * ig.dx[d];
            if (d == 2) { clo = ig.gamma_boost * (clo * ig.za - ig.zb); chi = ig.gamma_boost * (chi * ig.za - ig.zb); }   // :1021-1022
            // overlapsWith, and :1030-1048: a corner, an edge midpoint or the centre of the cell has density
            const double mid = (clo + chi) / 2.;
            const bool sample = (clo < ig.hi[d] && clo >= ig.lo[d]) || (mid < ig.hi[d] && mid >= ig.lo[d]) ||
                                (chi < ig.hi[d] && chi >= ig.lo[d]);
            ok = ok && !(clo > ig.hi[d] || chi < ig.lo[d]) && sample;
            pos[d] = ig.corner[d] + (iv[d] + r[d]) * ig.dx[d];                        // getCellCoords
            ok = ok && pos[d] > ig.blo[d] && pos[d] < ig.bhi[d];                      // tile_realbox.contains
            const double lab = d == 2 ? ig.gamma_boost * (pos[d] * ig.za - ig.zb) : pos[d];   // z0 / z0_lab (:1181, :1212)
            ok = ok && lab < ig.hi[d] && lab >= ig.lo[d];                             // insideBounds
        }
    }
    const unsigned long long mask = __ballot(ok);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader);
    if (!ok) return;
    const long slot = (long)(base + __popcll(mask & ((1ULL << lane) - 1ULL)));
    if (slot >= dst.np) return;   // the host sees count > room and reports it
    dst.x[slot] = pos[0]; dst.y[slot] = pos[1]; dst.z[slot] = pos[2];
    double u[3] = {ig.u[0], ig.u[1], ig.u[2]};
    if (ig.thermal) {
        double n[3];
        // lattice coordinate = cell * ppc + sub-point: (pos - origin) / dx * ppc = integer + 1/2, robust to round-off
        normal3(ig.seed, (int)floor((pos[0] - ig.origin[0]) / ig.dx[0] * ig.ppc[0]),
                (int)floor((pos[1] - ig.origin[1]) / ig.dx[1] * ig.ppc[1]),
                (int)floor((pos[2] - ig.origin[2]) / ig.dx[2] * ig.ppc[2]), n);
        u[0] += ig.uth[0] * n[0]; u[1] += ig.uth[1] * n[1]; u[2] += ig.uth[2] * n[2];
    }
    double dens = ig.density;
    if (ig.gamma_boost > 1.0) {   // :1232-1246 Lorentz transform of the lab-frame density and momentum
        const double gamma_lab = sqrt(1.0 + (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
        const double betaz_lab = u[2] / gamma_lab;
        dens = ig.gamma_boost * dens * (1.0 - ig.beta_boost * betaz_lab);
        u[2] = ig.gamma_boost * (u[2] - ig.beta_boost * gamma_lab);
    }
    dst.w[slot] = dens * ig.scale_fac;
    dst.ux[slot] = u[0] * PhysConst::c; dst.uy[slot] = u[1] * PhysConst::c; dst.uz[slot] = u[2] * PhysConst::c;
    if (dst.id) dst.id[slot] = 0;
}

// ---- particle walls: WarpXParticleContainer::ApplyBoundaryConditions -------------------------------
struct WallGeom {
    double lo[3], hi[3];
    int bc_lo[3], bc_hi[3];   // WXA_PBOUNDARY_*
};

// apply_boundary / apply_boundaries (Source/Particles/ParticleBoundaries_K.H:20-175): beyond a reflecting
// wall the position is mirrored and the normal momentum flips; beyond an absorbing wall the particle is
// lost = retired in place (weight and momentum 0, position clamped into the domain, idcpu marked).
__global__ void __launch_bounds__(256)
particle_walls_kernel(PV p, WallGeom wg, unsigned* __restrict__ n_lost) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.np) return;
    double x[3] = {p.x[ip], p.y[ip], p.z[ip]};
    if (p.id[ip] == WXA_IDCPU_RETIRED) {
        // A retired particle stays in the tile until the next sort and is still pushed (weight 0: it deposits
        // nothing): park it inside the domain again and take its momentum away, every step, so that it can never
        // drift out of the range the field arrays cover -- however long the sort is away
        bool out = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double c = fmin(fmax(x[d], wg.lo[d]), nextafter(wg.hi[d], wg.lo[d]));
            out = out || c != x[d];
            x[d] = c;
        }
        if (out) { p.x[ip] = x[0]; p.y[ip] = x[1]; p.z[ip] = x[2]; }
        p.ux[ip] = 0.0; p.uy[ip] = 0.0; p.uz[ip] = 0.0;
        return;
    }
    bool lost = false, flip[3] = {false, false, false}, moved = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (x[d] < wg.lo[d]) {
            if (wg.bc_lo[d] == WXA_PBOUNDARY_ABSORBING) lost = true;
            else if (wg.bc_lo[d] == WXA_PBOUNDARY_REFLECTING) { x[d] = 2 * wg.lo[d] - x[d]; flip[d] = true; moved = true; }
        } else if (x[d] > wg.hi[d]) {
            if (wg.bc_hi[d] == WXA_PBOUNDARY_ABSORBING) lost = true;
            else if (wg.bc_hi[d] == WXA_PBOUNDARY_REFLECTING) { x[d] = 2 * wg.hi[d] - x[d]; flip[d] = true; moved = true; }
        }
    }
    if (lost) {
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = fmin(fmax(x[d], wg.lo[d]), nextafter(wg.hi[d], wg.lo[d]));
        p.x[ip] = x[0]; p.y[ip] = x[1]; p.z[ip] = x[2];
        p.w[ip] = 0.0; p.ux[ip] = 0.0; p.uy[ip] = 0.0; p.uz[ip] = 0.0;
        p.id[ip] = WXA_IDCPU_RETIRED;
        atomicAdd(n_lost, 1u);
    } else if (moved) {
        if (flip[0]) { p.x[ip] = x[0]; p.ux[ip] = -p.ux[ip]; }
        if (flip[1]) { p.y[ip] = x[1]; p.uy[ip] = -p.uy[ip]; }
        if (flip[2]) { p.z[ip] = x[2]; p.uz[ip] = -p.uz[ip]; }
    }
}

// ---- Redistribute without moving the tile (see include/warpx_amd.h) --------------------
struct ClassifyGeom {
    double plo[3], phi[3], blo[3], bhi[3];
    int periodic[3], split[3];
};

// DEST = false: 6 lists, by the first split direction in which the particle is outside (before the wrap);
// DEST = true: 27 lists, by the offset of the destination brick (wxa_wrap_and_classify_dest)
template <bool DEST>
__global__ void __launch_bounds__(256)
wrap_classify_kernel(double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                     const uint64_t* __restrict__ id, long first, long count, ClassifyGeom cg,
                     int* __restrict__ lists, long cap, unsigned* __restrict__ counts) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const long ip = first + t;
    double v[3] = {x[ip], y[ip], z[ip]};
    int code = -1;
    if constexpr (DEST) {
        int o[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (cg.split[d]) o[d] = v[d] < cg.blo[d] ? -1 : (v[d] >= cg.bhi[d] ? 1 : 0);
        code = (o[0] + 1) + 3 * (o[1] + 1) + 9 * (o[2] + 1);
        if (code == 13) code = -1;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (cg.split[d] && code < 0) code = v[d] < cg.blo[d] ? 2 * d : (v[d] >= cg.bhi[d] ? 2 * d + 1 : -1);
    }
    if (code >= 0 && id[ip] == WXA_IDCPU_RETIRED) {
        // A retired particle is still pushed until the next sort drops it (weight 0: it deposits zeros).  It was
        // parked on the brick's side of the face it left through, so the push can carry it across again; if
        // that face is the domain boundary the periodic wrap would send it a whole domain away, where its
        // stencils lie outside this brick's arrays.  Park it again instead.
        code = -1;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (cg.split[d]) v[d] = fmin(fmax(v[d], cg.blo[d]), nextafter(cg.bhi[d], cg.blo[d]));
        if (v[0] != x[ip]) x[ip] = v[0];
        if (v[1] != y[ip]) y[ip] = v[1];
        if (v[2] != z[ip]) z[ip] = v[2];
    }
    if (cg.periodic[0]) { const double w = wrap_periodic(v[0], cg.plo[0], cg.phi[0]); if (w != v[0]) x[ip] = w; }
    if (cg.periodic[1]) { const double w = wrap_periodic(v[1], cg.plo[1], cg.phi[1]); if (w != v[1]) y[ip] = w; }
    if (cg.periodic[2]) { const double w = wrap_periodic(v[2], cg.plo[2], cg.phi[2]); if (w != v[2]) z[ip] = w; }
    if (code >= 0) {
        const unsigned slot = atomicAdd(&counts[code], 1u);
        if ((long)slot < cap) lists[code * cap + slot] = (int)ip;
    }
}

__global__ void __launch_bounds__(256)
pack_leavers_kernel(PV p, const int* __restrict__ list, long n, double* __restrict__ msg, long row_len,
                    long offset, int retire, ClassifyGeom cg) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long ip = list[t];
    double* m = msg + offset + t;
    m[0 * row_len] = p.x[ip]; m[1 * row_len] = p.y[ip]; m[2 * row_len] = p.z[ip]; m[3 * row_len] = p.w[ip];
    m[4 * row_len] = p.ux[ip]; m[5 * row_len] = p.uy[ip]; m[6 * row_len] = p.uz[ip];
    reinterpret_cast<uint64_t*>(m)[7 * row_len] = p.id[ip];
    if (retire) {
        // inert from here on: deposits exact zeros, stays inside the brick (and inside the reach of
        // its tile), and is dropped by the next sort
        double* pos[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double v = pos[d][ip];
            const double inside = fmin(fmax(v, cg.blo[d]), nextafter(cg.bhi[d], cg.blo[d]));
            if (inside != v) pos[d][ip] = inside;
        }
        p.w[ip] = 0.0; p.ux[ip] = 0.0; p.uy[ip] = 0.0; p.uz[ip] = 0.0;
        p.id[ip] = WXA_IDCPU_RETIRED;
    }
}

template <int PUSHER, bool MOVE>
static wxa_status launch_gather_push(const PV& pv, const wxa_field_view E[3], const wxa_field_view B[3],
                                     const Geom& g, double q, double m, double dt, int order, int galerkin,
                                     const ExtEB& ext, const PushSort& hook, hipStream_t st) {
    const DevF ex = make_devf(E[0]), ey = make_devf(E[1]), ez = make_devf(E[2]);
    const DevF bx = make_devf(B[0]), by = make_devf(B[1]), bz = make_devf(B[2]);
    const dim3 grid(blocks_for(pv.np)), block(256);
#define WXA_GP(O, G)                                                                              \
    hipLaunchKernelGGL((gather_push_kernel<O, G, PUSHER, MOVE>), grid, block, 0, st, pv, ex, ey, ez, bx, by, \
                       bz, g, q, m, dt, ext, hook)
    if (galerkin) {
        if (order == 1) WXA_GP(1, 1); else if (order == 2) WXA_GP(2, 1); else if (order == 3) WXA_GP(3, 1); else WXA_GP(4, 1);
    } else {
        if (order == 1) WXA_GP(1, 0); else if (order == 2) WXA_GP(2, 0); else if (order == 3) WXA_GP(3, 0); else WXA_GP(4, 0);
    }
#undef WXA_GP
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// per-particle external fields of the push that follows (none unless the workspace carries a lens)
static wxa_status evaluate_particle_fields(const wxa_particle_view* p, wxa_workspace* ws, hipStream_t st) {
    if (!ws) return WXA_OK;
    ws->ext_pp_stride = 0;
    if (ws->lens_n <= 0 || p->np == 0) return WXA_OK;
    wxa_status rc;
    if ((rc = ws->ext_pp.reserve(sizeof(double) * 4 * (size_t)p->np)) != WXA_OK) return rc;
    const PV pv = make_pv(*p);
    hipLaunchKernelGGL(lens_fields_kernel, dim3(blocks_for(pv.np)), dim3(256), 0, st, pv, lens_of(ws), (double*)ws->ext_pp.p,
                       (long)p->np);
    WXA_LAUNCH_CHECK();
    ws->ext_pp_stride = p->np;
    return WXA_OK;
}

static wxa_status check_gather_args(const wxa_particle_view* p, const wxa_field_view E[3],
                                    const wxa_field_view B[3], const wxa_grid_geom* geom, int order,
                                    int galerkin, int pusher) {
    WXA_REQUIRE(pv_ok(p), "bad particle view");
    WXA_REQUIRE(E && B && geom, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(E[c]) && view_ok(B[c]), "bad field view");
    WXA_REQUIRE(order >= 1 && order <= 4, "particle shape order must be 1..4");
    WXA_REQUIRE(galerkin == 0 || galerkin == 1, "galerkin must be 0 or 1");
    WXA_REQUIRE(pusher >= WXA_PUSHER_BORIS && pusher <= WXA_PUSHER_BORIS_RR,
                "pusher must be Boris, Vay, Higuera-Cary or Boris with radiation reaction");
    if (!yee_E(E) || !yee_B(B)) {
        set_last_error("gather: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    return WXA_OK;
}

}  // namespace wxa

using namespace wxa;

extern "C" {

// the global-memory kernel on `rest` with the external fields `ext`
static wxa_status gather_push_global(const wxa_particle_view& rest, const wxa_field_view E[3], const wxa_field_view B[3],
                                     const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                                     int pusher, int move, const ExtEB& ext, const PushSort& hook, hipStream_t st) {
    const PV pv = make_pv(rest);
    const Geom g = make_geom(*geom);
    if (pusher == WXA_PUSHER_BORIS) {
        if (move) return launch_gather_push<WXA_PUSHER_BORIS, true>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
        return launch_gather_push<WXA_PUSHER_BORIS, false>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
    }
    if (pusher == WXA_PUSHER_VAY) {
        if (move) return launch_gather_push<WXA_PUSHER_VAY, true>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
        return launch_gather_push<WXA_PUSHER_VAY, false>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
    }
    if (pusher == WXA_PUSHER_HC) {
        if (move) return launch_gather_push<WXA_PUSHER_HC, true>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
        return launch_gather_push<WXA_PUSHER_HC, false>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
    }
    if (move) return launch_gather_push<WXA_PUSHER_BORIS_RR, true>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
    return launch_gather_push<WXA_PUSHER_BORIS_RR, false>(pv, E, B, g, q, m, dt, order, galerkin, ext, hook, st);
}

wxa_status wxa_gather_push_ws(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                              const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                              int pusher, int move, wxa_workspace* ws, void* stream) {
    wxa_status rc = check_gather_args(p, E, B, geom, order, galerkin, pusher);
    if (rc != WXA_OK) return rc;
    if (p->np == 0) return WXA_OK;
    if ((rc = evaluate_particle_fields(p, ws, (hipStream_t)stream)) != WXA_OK) return rc;
    wxa_particle_view rest = *p;
    int64_t first = 0;
    if (gather_tile_available(ws, p)) {   // orders 1 .. 4 (4 since round 6: a tile of 13^3 / 14^3 staged points)
        // sorted part on the LDS tiles; particles appended since the sort (arrivals from the
        // neighbouring bricks) take the global-memory kernel below
        wxa_particle_view head = *p;
        head.np = ws->sorted_np;
        if (head.np > 0 &&
            (rc = gather_push_tiled(&head, E, B, geom, q, m, dt, order, galerkin, pusher, move != 0, ws,
                                    (hipStream_t)stream)) != WXA_OK)
            return rc;
        rest = tail_view(*p, ws->sorted_np);
        first = ws->sorted_np;
        if (rest.np == 0) return WXA_OK;
    }
    return gather_push_global(rest, E, B, geom, q, m, dt, order, galerkin, pusher, move, ext_of(ws, first),
                              make_push_sort(ws, first, move != 0), (hipStream_t)stream);
}

wxa_status wxa_gather_push_part(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                                const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                                int pusher, wxa_workspace* ws, int part, void* stream) {
    wxa_status rc = check_gather_args(p, E, B, geom, order, galerkin, pusher);
    if (rc != WXA_OK) return rc;
    WXA_REQUIRE(part == WXA_PART_INTERIOR || part == WXA_PART_REST, "part must be WXA_PART_INTERIOR or WXA_PART_REST");
    if (p->np == 0) return WXA_OK;
    if ((rc = evaluate_particle_fields(p, ws, (hipStream_t)stream)) != WXA_OK) return rc;
    if (!gather_tile_available(ws, p)) {   // no tiles: the interior part is empty, the rest is everything
        if (part == WXA_PART_INTERIOR) return WXA_OK;
        return gather_push_global(*p, E, B, geom, q, m, dt, order, galerkin, pusher, 1, ext_of(ws), make_push_sort(ws, 0, true),
                                  (hipStream_t)stream);
    }
    wxa_particle_view head = *p;
    head.np = ws->sorted_np;
    if (head.np > 0 &&
        (rc = gather_push_tiled_part(&head, E, B, geom, q, m, dt, order, galerkin, pusher, part, ws,
                                     (hipStream_t)stream)) != WXA_OK)
        return rc;
    if (part == WXA_PART_INTERIOR) return WXA_OK;
    const wxa_particle_view rest = tail_view(*p, ws->sorted_np);   // arrivals since the sort may sit anywhere
    if (rest.np == 0) return WXA_OK;
    return gather_push_global(rest, E, B, geom, q, m, dt, order, galerkin, pusher, 1, ext_of(ws, ws->sorted_np),
                              make_push_sort(ws, ws->sorted_np, true), (hipStream_t)stream);
}

wxa_status wxa_gather_push(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                           const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                           int pusher, void* stream) {
    return wxa_gather_push_ws(p, E, B, geom, q, m, dt, order, galerkin, pusher, 1, nullptr, stream);
}

wxa_status wxa_push_p(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                      const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                      int pusher, void* stream) {
    return wxa_gather_push_ws(p, E, B, geom, q, m, dt, order, galerkin, pusher, 0, nullptr, stream);
}

wxa_status wxa_deposit_current(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* geom,
                               double q, double dt, double relative_time, int order, int algo,
                               wxa_workspace* ws, void* stream) {
    WXA_REQUIRE(pv_ok(p), "bad particle view");
    WXA_REQUIRE(J && geom, "null argument");
    for (int c = 0; c < 3; ++c) WXA_REQUIRE(view_ok(J[c]), "bad field view");
    WXA_REQUIRE(order >= 1 && order <= 4, "particle shape order must be 1..4");
    WXA_REQUIRE(algo == WXA_DEPOSIT_ESIRKEPOV || algo == WXA_DEPOSIT_DIRECT, "unknown deposition algorithm");
    WXA_REQUIRE(dt > 0.0 || algo == WXA_DEPOSIT_DIRECT, "dt must be positive");
    if (!yee_E(J)) {
        set_last_error("deposit_current: only the Yee staggering is supported");
        return WXA_ERR_UNSUPPORTED;
    }
    if (p->np == 0) return WXA_OK;
    wxa_particle_view rest = *p;
    if (ws && deposit_tile_available(ws, p)) {   // orders 1 .. 4 (4 since round 6: the tile's points are exactly the quartic stencil's reach)
        wxa_particle_view head = *p;
        head.np = ws->sorted_np;
        wxa_status rc;
        if (head.np > 0 && (rc = deposit_current_tiled(&head, J, geom, q, dt, relative_time, order, algo, ws,
                                                       (hipStream_t)stream)) != WXA_OK)
            return rc;
        rest = tail_view(*p, ws->sorted_np);   // arrivals since the sort: global atomics
        if (rest.np == 0) return WXA_OK;
    }
    const PV pv = make_pv(rest);
    const Geom g = make_geom(*geom);
    const DevF jx = make_devf(J[0]), jy = make_devf(J[1]), jz = make_devf(J[2]);
    const dim3 grid(blocks_for(pv.np)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (algo == WXA_DEPOSIT_ESIRKEPOV) {
        const EsirkepovStep es = make_esirkepov_step(g, dt, relative_time);
        if (order == 1) hipLaunchKernelGGL(deposit_esirkepov_global_kernel<1>, grid, block, 0, st, pv, jx, jy, jz, g, q, es);
        else if (order == 2) hipLaunchKernelGGL(deposit_esirkepov_global_kernel<2>, grid, block, 0, st, pv, jx, jy, jz, g, q, es);
        else if (order == 3) hipLaunchKernelGGL(deposit_esirkepov_global_kernel<3>, grid, block, 0, st, pv, jx, jy, jz, g, q, es);
        else hipLaunchKernelGGL(deposit_esirkepov_global_kernel<4>, grid, block, 0, st, pv, jx, jy, jz, g, q, es);
    } else {
        if (order == 1) hipLaunchKernelGGL(deposit_direct_global_kernel<1>, grid, block, 0, st, pv, jx, jy, jz, g, q, relative_time);
        else if (order == 2) hipLaunchKernelGGL(deposit_direct_global_kernel<2>, grid, block, 0, st, pv, jx, jy, jz, g, q, relative_time);
        else if (order == 3) hipLaunchKernelGGL(deposit_direct_global_kernel<3>, grid, block, 0, st, pv, jx, jy, jz, g, q, relative_time);
        else hipLaunchKernelGGL(deposit_direct_global_kernel<4>, grid, block, 0, st, pv, jx, jy, jz, g, q, relative_time);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_deposit_charge(const wxa_particle_view* p, const wxa_field_view* rho, const wxa_grid_geom* geom,
                              double q, int order, void* stream) {
    WXA_REQUIRE(pv_ok(p), "bad particle view");
    WXA_REQUIRE(rho && geom && view_ok(*rho), "bad argument");
    WXA_REQUIRE(order >= 1 && order <= 4, "particle shape order must be 1..4");
    if (p->np == 0) return WXA_OK;
    const PV pv = make_pv(*p);
    const Geom g = make_geom(*geom);
    const DevF r = make_devf(*rho);
    const dim3 grid(blocks_for(pv.np)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (order == 1) hipLaunchKernelGGL(deposit_charge_kernel<1>, grid, block, 0, st, pv, r, rho->stag[0], rho->stag[1], rho->stag[2], g, q);
    else if (order == 2) hipLaunchKernelGGL(deposit_charge_kernel<2>, grid, block, 0, st, pv, r, rho->stag[0], rho->stag[1], rho->stag[2], g, q);
    else if (order == 3) hipLaunchKernelGGL(deposit_charge_kernel<3>, grid, block, 0, st, pv, r, rho->stag[0], rho->stag[1], rho->stag[2], g, q);
    else hipLaunchKernelGGL(deposit_charge_kernel<4>, grid, block, 0, st, pv, r, rho->stag[0], rho->stag[1], rho->stag[2], g, q);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_enforce_periodic(const wxa_particle_view* p, const double plo[3], const double phi[3],
                                const int periodic[3], void* stream) {
    WXA_REQUIRE(pv_ok(p) && plo && phi && periodic, "bad argument");
    if (p->np == 0) return WXA_OK;
    PeriodicBox pb;
    bool any = false;
    for (int d = 0; d < 3; ++d) {
        pb.plo[d] = plo[d]; pb.phi[d] = phi[d]; pb.on[d] = periodic[d] ? 1 : 0;
        if (periodic[d]) { WXA_REQUIRE(phi[d] > plo[d], "empty domain"); any = true; }
    }
    if (!any) return WXA_OK;
    hipLaunchKernelGGL(enforce_periodic_kernel, dim3(blocks_for(p->np)), dim3(256), 0, (hipStream_t)stream, p->x,
                       p->y, p->z, (long)p->np, pb);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// The periodic wrap restricted to the particles that can need it: those of the tiles of the last cell sort that touch
// a periodic face of the sorted box (= the domain on a single brick), plus the particles appended since.  A particle
// moves less than one cell per step, so while fewer steps than a tile is wide have passed since the sort, a particle
// of an interior tile cannot have left the domain.  Same arithmetic as wxa_enforce_periodic on 18 % of the particles
// (256^3 in tiles of 8^3).
constexpr int EPT_THREADS = 256, EPT_SPLIT = 4;   // four workgroups per tile, x 4 particles per lane: a quarter of a tile of 8 per cell in one pass (per launch, 256^3: 0.300 ms with one workgroup per tile, 0.342 with one of 1024 lanes, 0.172 with four, 0.253 with eight)
__global__ void __launch_bounds__(EPT_THREADS)
enforce_periodic_tiles_kernel(double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                              const int* __restrict__ offsets, int nt0, int nt1, int nt2, int nc0, int nc1, int nc2,
                              PeriodicBox pb) {
    constexpr int T = WXA_TILE;
    const int tile = blockIdx.x / EPT_SPLIT, part = blockIdx.x % EPT_SPLIT;
    const int ti = tile % nt0, tj = (tile / nt0) % nt1, tk = tile / (nt0 * nt1);
    // a tile whose cells lie within a tile width of a periodic face (the last tile of a direction may be partial)
    const bool face = (pb.on[0] && (ti == 0 || (ti + 2) * T > nc0)) || (pb.on[1] && (tj == 0 || (tj + 2) * T > nc1)) ||
                      (pb.on[2] && (tk == 0 || (tk + 2) * T > nc2));
    if (!face) return;
    constexpr int TC = WXA_TILE * WXA_TILE * WXA_TILE;
    const int t_start = offsets[(long)tile * TC], t_end = offsets[(long)(tile + 1) * TC];
    const int per = (t_end - t_start + EPT_SPLIT - 1) / EPT_SPLIT;
    const int start = t_start + part * per, end = min(start + per, t_end);
    // four particles per lane and pass, all their loads in flight before the first (conditional) store: written as one
    // load - test - store after the other, the kernel had one load in flight per lane (0.49 ms for 18 % of the particles)
    constexpr int U = 4;
    for (int base = start + (int)threadIdx.x; base < end; base += EPT_THREADS * U) {
        double v[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ip = base + u * EPT_THREADS;
            const bool in = ip < end;
            v[u][0] = in && pb.on[0] ? x[ip] : 0.0;
            v[u][1] = in && pb.on[1] ? y[ip] : 0.0;
            v[u][2] = in && pb.on[2] ? z[ip] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ip = base + u * EPT_THREADS;
            if (ip >= end) continue;
            if (pb.on[0]) { const double w = wrap_periodic(v[u][0], pb.plo[0], pb.phi[0]); if (w != v[u][0]) x[ip] = w; }
            if (pb.on[1]) { const double w = wrap_periodic(v[u][1], pb.plo[1], pb.phi[1]); if (w != v[u][1]) y[ip] = w; }
            if (pb.on[2]) { const double w = wrap_periodic(v[u][2], pb.plo[2], pb.phi[2]); if (w != v[u][2]) z[ip] = w; }
        }
    }
}

wxa_status wxa_enforce_periodic_sorted(const wxa_particle_view* p, const double plo[3], const double phi[3],
                                       const int periodic[3], wxa_workspace* ws, int32_t steps_since_sort,
                                       void* stream) {
    WXA_REQUIRE(pv_ok(p) && plo && phi && periodic, "bad argument");
    // no usable sort, or the drift since it may exceed a tile: the plain pass over everything
    if (!ws || !ws->sorted_valid || ws->sorted_x != p->x || ws->sorted_np > p->np || steps_since_sort < 0 ||
        steps_since_sort > WXA_TILE - 2)
        return wxa_enforce_periodic(p, plo, phi, periodic, stream);
    if (p->np == 0) return WXA_OK;
    PeriodicBox pb;
    bool any = false;
    for (int d = 0; d < 3; ++d) {
        pb.plo[d] = plo[d]; pb.phi[d] = phi[d]; pb.on[d] = periodic[d] ? 1 : 0;
        if (periodic[d]) { WXA_REQUIRE(phi[d] > plo[d], "empty domain"); any = true; }
    }
    if (!any) return WXA_OK;
    const int nt0 = (ws->sort_nc[0] + WXA_TILE - 1) / WXA_TILE, nt1 = (ws->sort_nc[1] + WXA_TILE - 1) / WXA_TILE,
              nt2 = (ws->sort_nc[2] + WXA_TILE - 1) / WXA_TILE;
    hipLaunchKernelGGL(enforce_periodic_tiles_kernel, dim3((unsigned)(nt0 * nt1 * nt2) * EPT_SPLIT), dim3(EPT_THREADS), 0, (hipStream_t)stream,
                       p->x, p->y, p->z, (const int*)ws->offsets.p, nt0, nt1, nt2, ws->sort_nc[0], ws->sort_nc[1],
                       ws->sort_nc[2], pb);
    if (p->np > ws->sorted_np) {   // appended since the sort
        const wxa_particle_view tail = tail_view(*p, ws->sorted_np);
        hipLaunchKernelGGL(enforce_periodic_kernel, dim3(blocks_for(tail.np)), dim3(256), 0, (hipStream_t)stream,
                           tail.x, tail.y, tail.z, (long)tail.np, pb);
    }
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_sort_particles_by_cell(const wxa_particle_view* src, const wxa_particle_view* dst,
                                      const double plo[3], const double dinv[3], const int32_t cell_lo[3],
                                      const int32_t ncell[3], wxa_workspace* ws, void* stream) {
    WXA_REQUIRE(pv_ok(src) && pv_ok(dst) && plo && dinv && cell_lo && ncell && ws, "bad argument");
    WXA_REQUIRE(src->np == dst->np, "src/dst particle counts differ");
    WXA_REQUIRE(src->x != dst->x, "sort is out of place");
    WXA_REQUIRE(ncell[0] > 0 && ncell[1] > 0 && ncell[2] > 0, "empty cell box");
    const long ncells = (long)((ncell[0] + WXA_TILE - 1) / WXA_TILE) * ((ncell[1] + WXA_TILE - 1) / WXA_TILE) *
                        ((ncell[2] + WXA_TILE - 1) / WXA_TILE) * (WXA_TILE * WXA_TILE * WXA_TILE);
    WXA_REQUIRE(ncells > 0 && ncells < (1L << 31) - 2 && src->np < (1L << 31) - 2, "sizes exceed 32-bit sort keys");
    hipStream_t st = (hipStream_t)stream;
    ws->sorted_valid = false;
    ws->ps.pending = false;   // a record of wxa_push_sort_begin(COUNT) indexes the order this sort replaces
    if (src->np == 0) return WXA_OK;
    wxa_status rc;
    if ((rc = ws->rank.reserve(sizeof(int) * src->np)) != WXA_OK) return rc;   // (no key array: the scatter works the keys out again)
    // bins: the cells, the retired particles, and one closing entry for the scan
    if ((rc = ws->hist.reserve(sizeof(int) * (ncells + 2))) != WXA_OK) return rc;
    if ((rc = ws->offsets.reserve(sizeof(int) * (ncells + 2))) != WXA_OK) return rc;
    int* rank = (int*)ws->rank.p;
    int* hist = (int*)ws->hist.p; int* offsets = (int*)ws->offsets.p;
    WXA_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(int) * (ncells + 2), st));
    SortGeom sg;
    for (int d = 0; d < 3; ++d) {
        // physical lower corner of the brick's cell box; cells are numbered from cell_lo
        sg.plo[d] = plo[d];
        sg.dinv[d] = dinv[d];
        sg.nc[d] = ncell[d];
    }
    sg.retired_bin = (int)ncells;
    (void)cell_lo;
    const PV s = make_pv(*src), d = make_pv(*dst);
    hipLaunchKernelGGL(sort_count_kernel, dim3(blocks_for(s.np)), dim3(256), 0, st, s.x, s.y, s.z, s.id, s.np, sg,
                       (int*)nullptr, rank, hist);
    size_t tmp_bytes = 0;
    WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, hist, offsets, (int)(ncells + 2), st));
    if ((rc = ws->scan_tmp.reserve(tmp_bytes)) != WXA_OK) return rc;
    WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ws->scan_tmp.p, tmp_bytes, hist, offsets, (int)(ncells + 2), st));
    // the scatter works the keys out again from the positions it loads (round 4) and stages a chunk's near-stayers in LDS
    // in destination order, so that it writes whole lines.  Window shapes timed at 256^3 x 8 ppc (Redistribute per step;
    // the plain scatter -- one lane per particle, 8-byte writes wherever they fall -- 1.47): 8 x 512 lanes + 512 margin
    // 1.24, 4 x 512 + 512 1.29, 8 x 512 + 1024 1.25, 16 x 512 + 512 1.38
    hipLaunchKernelGGL((sort_scatter_window_kernel<8, 512, true>), dim3(blocks_for(s.np, SW_THREADS * 8)), dim3(SW_THREADS), 0,
                       st, s, d, (const int*)nullptr, rank, offsets, sg);
    WXA_LAUNCH_CHECK();
    ws->sorted_valid = true;
    ws->sorted_np = src->np;   // wxa_sort_live_count lowers it to the live count
    ws->sorted_bins = ncells;
    ws->sorted_x = dst->x;
    for (int e = 0; e < 3; ++e) {
        ws->sort_nc[e] = ncell[e];
        ws->sort_cell_lo[e] = cell_lo[e];
        ws->sort_plo[e] = plo[e];
        ws->sort_dinv[e] = dinv[e];
    }
    return WXA_OK;
}

wxa_status wxa_partition_particles(const wxa_particle_view* src, const wxa_particle_view* dst, int dim,
                                   double lo, double hi, int64_t counts[3], wxa_workspace* ws, void* stream) {
    WXA_REQUIRE(pv_ok(src) && pv_ok(dst) && counts && ws, "bad argument");
    WXA_REQUIRE(dim >= 0 && dim < 3, "dim must be 0..2");
    WXA_REQUIRE(src->np == dst->np && (src->np == 0 || src->x != dst->x), "partition is out of place, equal sizes");
    WXA_REQUIRE(src->np < (1L << 31) - 2, "tile too large for 32-bit scan");
    hipStream_t st = (hipStream_t)stream;
    counts[0] = counts[1] = counts[2] = 0;
    ws->sorted_valid = false;
    ws->ps.pending = false;
    if (src->np == 0) return WXA_OK;
    wxa_status rc;
    if ((rc = ws->cell.reserve(sizeof(int) * (src->np + 1))) != WXA_OK) return rc;
    if ((rc = ws->rank.reserve(sizeof(int) * (src->np + 1))) != WXA_OK) return rc;
    if ((rc = ws->hist.reserve(sizeof(unsigned long long) * 8)) != WXA_OK) return rc;
    int* stay = (int*)ws->cell.p; int* scan = (int*)ws->rank.p;
    unsigned long long* ctr = (unsigned long long*)ws->hist.p;
    WXA_HIP_CHECK(hipMemsetAsync(ctr, 0, sizeof(unsigned long long) * 8, st));
    const PV s = make_pv(*src), d = make_pv(*dst);
    const double* pos = dim == 0 ? s.x : (dim == 1 ? s.y : s.z);
    hipLaunchKernelGGL(partition_flag_kernel, dim3(blocks_for(s.np)), dim3(256), 0, st, pos, s.np, lo, hi, stay, ctr);
    size_t tmp_bytes = 0;
    WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, stay, scan, (int)s.np, st));
    if ((rc = ws->scan_tmp.reserve(tmp_bytes)) != WXA_OK) return rc;
    WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ws->scan_tmp.p, tmp_bytes, stay, scan, (int)s.np, st));
    unsigned long long h[3] = {0, 0, 0};
    WXA_HIP_CHECK(hipMemcpyAsync(h, ctr, sizeof(h), hipMemcpyDeviceToHost, st));
    WXA_HIP_CHECK(hipStreamSynchronize(st));
    const long nminus = (long)h[1], nplus = (long)h[2], nstay = s.np - nminus - nplus;
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(blocks_for(s.np)), dim3(256), 0, st, s, d, pos, lo, hi, scan,
                       nstay, nminus, ctr + 3);
    WXA_LAUNCH_CHECK();
    counts[0] = nstay; counts[1] = nminus; counts[2] = nplus;
    return WXA_OK;
}

}  // extern "C"

template <bool DEST>
static wxa_status wrap_and_classify_impl(const wxa_particle_view* p, int64_t first, int64_t count, const double prob_lo[3],
                                         const double prob_hi[3], const int periodic[3], const double brick_lo[3],
                                         const double brick_hi[3], const int split[3], int32_t* lists, int64_t capacity,
                                         int64_t* counts, wxa_workspace* ws, void* stream) {
    constexpr int NL = DEST ? 27 : 6;
    WXA_REQUIRE(pv_ok(p) && prob_lo && prob_hi && periodic && brick_lo && brick_hi && split && counts && ws,
                "bad argument");
    WXA_REQUIRE(first >= 0 && count >= 0 && first + count <= p->np, "range outside the tile");
    WXA_REQUIRE(capacity >= 0 && (capacity == 0 || lists), "null list storage");
    const bool any_split = split[0] || split[1] || split[2];
    WXA_REQUIRE(!any_split || p->idcpu, "idcpu is needed to recognise retired particles");
    for (int c = 0; c < NL; ++c) counts[c] = 0;
    if (count == 0) return WXA_OK;
    ClassifyGeom cg;
    for (int d = 0; d < 3; ++d) {
        cg.plo[d] = prob_lo[d]; cg.phi[d] = prob_hi[d]; cg.blo[d] = brick_lo[d]; cg.bhi[d] = brick_hi[d];
        cg.periodic[d] = periodic[d] ? 1 : 0; cg.split[d] = split[d] ? 1 : 0;
        if (periodic[d]) WXA_REQUIRE(prob_hi[d] > prob_lo[d], "empty domain");
    }
    hipStream_t st = (hipStream_t)stream;
    wxa_status rc;
    if ((rc = ws->counters.reserve(512)) != WXA_OK) return rc;
    unsigned* dcount = (unsigned*)ws->counters.p + (DEST ? 64 : 32);   // words: 0 deposit, 16 gather, 32 classify, 48 walls, 56 injection, 64..90 destinations
    WXA_HIP_CHECK(hipMemsetAsync(dcount, 0, NL * sizeof(unsigned), st));
    hipLaunchKernelGGL(wrap_classify_kernel<DEST>, dim3(blocks_for(count)), dim3(256), 0, st, p->x, p->y, p->z, p->idcpu,
                       (long)first, (long)count, cg, lists, (long)capacity, dcount);
    WXA_LAUNCH_CHECK();
    if (!any_split) return WXA_OK;   // nothing can be listed: no need to wait
    unsigned h[NL];
    WXA_HIP_CHECK(hipMemcpyAsync(h, dcount, sizeof(h), hipMemcpyDeviceToHost, st));
    WXA_HIP_CHECK(hipStreamSynchronize(st));
    for (int c = 0; c < NL; ++c) counts[c] = h[c];
    return WXA_OK;
}

extern "C" {

wxa_status wxa_wrap_and_classify(const wxa_particle_view* p, int64_t first, int64_t count, const double prob_lo[3],
                                 const double prob_hi[3], const int periodic[3], const double brick_lo[3],
                                 const double brick_hi[3], const int split[3], int32_t* lists, int64_t capacity,
                                 int64_t counts[6], wxa_workspace* ws, void* stream) {
    return wrap_and_classify_impl<false>(p, first, count, prob_lo, prob_hi, periodic, brick_lo, brick_hi, split, lists,
                                         capacity, counts, ws, stream);
}

wxa_status wxa_wrap_and_classify_dest(const wxa_particle_view* p, int64_t first, int64_t count, const double prob_lo[3],
                                      const double prob_hi[3], const int periodic[3], const double brick_lo[3],
                                      const double brick_hi[3], const int split[3], int32_t* lists, int64_t capacity,
                                      int64_t counts[27], wxa_workspace* ws, void* stream) {
    return wrap_and_classify_impl<true>(p, first, count, prob_lo, prob_hi, periodic, brick_lo, brick_hi, split, lists,
                                        capacity, counts, ws, stream);
}

wxa_status wxa_pack_leavers(const wxa_particle_view* p, const int32_t* list, int64_t n, void* msg, int64_t row_len,
                            int64_t offset, int retire, const double brick_lo[3], const double brick_hi[3],
                            void* stream) {
    WXA_REQUIRE(pv_ok(p) && p->idcpu && brick_lo && brick_hi, "bad argument");
    WXA_REQUIRE(n >= 0 && (n == 0 || (list && msg)), "null list or message");
    WXA_REQUIRE(offset >= 0 && offset + n <= row_len, "list does not fit the message rows");
    if (n == 0) return WXA_OK;
    ClassifyGeom cg{};
    for (int d = 0; d < 3; ++d) { cg.blo[d] = brick_lo[d]; cg.bhi[d] = brick_hi[d]; }
    hipLaunchKernelGGL(pack_leavers_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, make_pv(*p),
                       list, (long)n, (double*)msg, (long)row_len, (long)offset, retire, cg);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_add_plasma(const wxa_particle_view* dst, const wxa_plasma_injector* inj, const double corner[3],
                          const int32_t ncells[3], const double dx[3], const double brick_lo[3],
                          const double brick_hi[3], const wxa_injected_momentum* mom, int64_t* n_added,
                          wxa_workspace* ws, void* stream) {
    WXA_REQUIRE(dst && inj && corner && ncells && dx && brick_lo && brick_hi && n_added && ws, "null argument");
    WXA_REQUIRE(dst->np >= 0 && (dst->np == 0 || (dst->x && dst->y && dst->z && dst->w && dst->ux && dst->uy && dst->uz)),
                "bad particle view");
    WXA_REQUIRE(inj->ppc[0] >= 1 && inj->ppc[1] >= 1 && inj->ppc[2] >= 1 && inj->density >= 0.0, "bad injector");
    *n_added = 0;
    InjectGeom ig;
    long npoints = (long)inj->ppc[0] * inj->ppc[1] * inj->ppc[2];
    for (int d = 0; d < 3; ++d) {
        WXA_REQUIRE(ncells[d] >= 0 && dx[d] > 0, "bad cell box");
        ig.corner[d] = corner[d]; ig.dx[d] = dx[d]; ig.blo[d] = brick_lo[d]; ig.bhi[d] = brick_hi[d];
        ig.lo[d] = inj->lo[d]; ig.hi[d] = inj->hi[d]; ig.nc[d] = ncells[d]; ig.ppc[d] = inj->ppc[d];
        ig.u[d] = mom ? mom->u_mean[d] : 0.0;
        ig.uth[d] = mom ? mom->u_th[d] : 0.0;
        ig.origin[d] = mom ? mom->origin[d] : 0.0;
        npoints *= ncells[d];
    }
    ig.seed = mom ? mom->seed : 0;
    ig.thermal = mom && (mom->u_th[0] != 0.0 || mom->u_th[1] != 0.0 || mom->u_th[2] != 0.0);
    if (npoints == 0 || !(inj->density > 0)) return WXA_OK;
    ig.density = inj->density;
    ig.scale_fac = dx[0] * dx[1] * dx[2] / (inj->ppc[0] * inj->ppc[1] * inj->ppc[2]);   // compute_scale_fac_volume
    ig.gamma_boost = inj->gamma_boost > 1.0 ? inj->gamma_boost : 1.0;
    ig.beta_boost = ig.gamma_boost > 1.0 ? std::sqrt(1.0 - 1.0 / std::pow(ig.gamma_boost, 2.0)) : 0.0;
    {
        const double gamma_bulk = std::sqrt(1.0 + (ig.u[0] * ig.u[0] + ig.u[1] * ig.u[1] + ig.u[2] * ig.u[2]));
        const double betaz_bulk = ig.u[2] / gamma_bulk;
        ig.za = 1.0 - ig.beta_boost * betaz_bulk;
        ig.zb = PhysConst::c * inj->t * (betaz_bulk - ig.beta_boost);
    }
    hipStream_t st = (hipStream_t)stream;
    wxa_status rc;
    if ((rc = ws->counters.reserve(512)) != WXA_OK) return rc;
    unsigned long long* dcount = (unsigned long long*)((unsigned*)ws->counters.p + 56);   // 0: deposit, 16: gather, 32: classify, 48: walls
    WXA_HIP_CHECK(hipMemsetAsync(dcount, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(add_plasma_kernel, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, st, make_pv(*dst), ig, npoints,
                       dcount);
    WXA_LAUNCH_CHECK();
    unsigned long long h = 0;
    WXA_HIP_CHECK(hipMemcpyAsync(&h, dcount, sizeof(h), hipMemcpyDeviceToHost, st));
    WXA_HIP_CHECK(hipStreamSynchronize(st));
    if ((int64_t)h > dst->np) {
        set_last_error("wxa_add_plasma: not enough room for the injected particles");
        return WXA_ERR_NOMEM;
    }
    *n_added = (int64_t)h;
    return WXA_OK;
}

wxa_status wxa_laser_push(const wxa_particle_view* p, const wxa_laser_push_params* c, double t, double dt,
                          void* stream) {
    WXA_REQUIRE(pv_ok(p) && c, "bad argument");
    WXA_REQUIRE(c->wavelength > 0 && c->waist > 0 && c->duration > 0, "laser: wavelength, waist and duration must be > 0");
    if (p->np == 0) return WXA_OK;
    using cplx = std::complex<double>;
    const cplx I(0, 1);
    const double k0 = 2. * M_PI / c->wavelength;
    const double inv_tau2 = 1. / (c->duration * c->duration);
    const double oscillation_phase = k0 * PhysConst::c * (t - c->t_peak);
    const cplx diffract_factor = 1. + I * c->focal_distance * 2. / (k0 * c->waist * c->waist);
    const cplx inv_complex_waist_2 = 1. / (c->waist * c->waist * diffract_factor);
    const cplx prefactor = c->e_max * std::exp(I * oscillation_phase) / diffract_factor;
    const cplx stcfactor = prefactor * std::exp(-cplx(inv_tau2 * (t - c->t_peak) * (t - c->t_peak), 0.0));
    LaserPushGeom lg;
    for (int d = 0; d < 3; ++d) { lg.position[d] = c->position[d]; lg.p_X[d] = c->p_X[d]; lg.p_Y[d] = c->p_Y[d]; }
    lg.mobility = c->mobility;
    lg.pre_re = stcfactor.real(); lg.pre_im = stcfactor.imag();
    lg.iw2_re = inv_complex_waist_2.real(); lg.iw2_im = inv_complex_waist_2.imag();
    lg.gamma_boost = c->gamma_boost > 1.0 ? c->gamma_boost : 1.0;
    const double beta_boost = lg.gamma_boost > 1.0 ? std::sqrt(1.0 - 1.0 / std::pow(lg.gamma_boost, 2.0)) : 0.0;
    for (int d = 0; d < 3; ++d) lg.drift[d] = lg.gamma_boost > 1.0 ? PhysConst::c * beta_boost * c->nvec[d] : 0.0;
    hipLaunchKernelGGL(laser_push_kernel, dim3(blocks_for(p->np)), dim3(256), 0, (hipStream_t)stream, make_pv(*p), lg, dt);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status wxa_apply_particle_boundaries(const wxa_particle_view* p, const double prob_lo[3],
                                         const double prob_hi[3], const int32_t bc_lo[3], const int32_t bc_hi[3],
                                         int64_t* n_lost, wxa_workspace* ws, void* stream) {
    WXA_REQUIRE(pv_ok(p) && prob_lo && prob_hi && bc_lo && bc_hi && ws, "bad argument");
    if (n_lost) *n_lost = 0;
    bool any = false;
    WallGeom wg;
    for (int d = 0; d < 3; ++d) {
        WXA_REQUIRE(prob_hi[d] > prob_lo[d], "empty domain");
        wg.lo[d] = prob_lo[d]; wg.hi[d] = prob_hi[d];
        wg.bc_lo[d] = bc_lo[d]; wg.bc_hi[d] = bc_hi[d];
        any = any || bc_lo[d] == WXA_PBOUNDARY_ABSORBING || bc_lo[d] == WXA_PBOUNDARY_REFLECTING ||
              bc_hi[d] == WXA_PBOUNDARY_ABSORBING || bc_hi[d] == WXA_PBOUNDARY_REFLECTING;
    }
    if (!any || p->np == 0) return WXA_OK;
    WXA_REQUIRE(p->idcpu, "idcpu is needed to retire absorbed particles");
    hipStream_t st = (hipStream_t)stream;
    wxa_status rc;
    if ((rc = ws->counters.reserve(512)) != WXA_OK) return rc;
    unsigned* dcount = (unsigned*)ws->counters.p + 48;
    WXA_HIP_CHECK(hipMemsetAsync(dcount, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(particle_walls_kernel, dim3(blocks_for(p->np)), dim3(256), 0, st, make_pv(*p), wg, dcount);
    WXA_LAUNCH_CHECK();
    if (n_lost) {
        unsigned h = 0;
        WXA_HIP_CHECK(hipMemcpyAsync(&h, dcount, sizeof(h), hipMemcpyDeviceToHost, st));
        WXA_HIP_CHECK(hipStreamSynchronize(st));
        *n_lost = h;
    }
    return WXA_OK;
}

// ---- the cell sort folded into PushPX (push_sort.hpp) ----------------------------------------------------------------
wxa_status wxa_push_sort_begin(wxa_workspace* ws, int32_t mode, const wxa_particle_view* p, const wxa_particle_view* dst,
                               const double plo[3], const double dinv[3], const int32_t cell_lo[3], const int32_t ncell[3],
                               const int32_t wrap[3], int32_t check_retired, double predict_dt, void* stream) {
    WXA_REQUIRE(ws && pv_ok(p), "bad argument");
    WXA_REQUIRE(mode == WXA_PUSH_SORT_COUNT || mode == WXA_PUSH_SORT_SCATTER || mode == (WXA_PUSH_SORT_COUNT | WXA_PUSH_SORT_SCATTER),
                "mode must be WXA_PUSH_SORT_COUNT, WXA_PUSH_SORT_SCATTER or both");
    auto& s = ws->ps;
    WXA_REQUIRE(s.armed == 0, "wxa_push_sort_begin without the wxa_push_sort_end of the previous one");
    WXA_REQUIRE(p->np < (1L << 31) - 2, "sizes exceed 32-bit sort keys");
    hipStream_t st = (hipStream_t)stream;
    wxa_status rc;
    if (mode & WXA_PUSH_SORT_SCATTER) {
        WXA_REQUIRE(s.pending && s.pending_x == p->x && s.pending_np <= p->np,
                    "no record of a COUNT on these particle arrays (or they shrank since)");
        WXA_REQUIRE(pv_ok(dst) && dst->np >= p->np && (p->np == 0 || dst->x != p->x), "the destination tile: out of place, at least as long");
        WXA_REQUIRE((p->idcpu == nullptr) == (dst->idcpu == nullptr), "both tiles with ids or both without");
        s.dst = *dst;
        s.appended = p->np - s.pending_np;
    }
    if (mode & WXA_PUSH_SORT_COUNT) {
        WXA_REQUIRE(plo && dinv && cell_lo && ncell && wrap, "null argument");
        WXA_REQUIRE(ncell[0] > 0 && ncell[1] > 0 && ncell[2] > 0, "empty cell box");
        const long ncells = (long)((ncell[0] + WXA_TILE - 1) / WXA_TILE) * ((ncell[1] + WXA_TILE - 1) / WXA_TILE) *
                            ((ncell[2] + WXA_TILE - 1) / WXA_TILE) * (WXA_TILE * WXA_TILE * WXA_TILE);
        WXA_REQUIRE(ncells > 0 && ncells < (1L << 31) - 2, "sizes exceed 32-bit sort keys");
        s.out = (mode & WXA_PUSH_SORT_SCATTER) ? 1 - s.in : s.in;   // a record nobody used is overwritten
        if (!(mode & WXA_PUSH_SORT_SCATTER)) s.pending = false;
        if ((rc = s.kr[s.out].reserve(sizeof(unsigned long long) * (size_t)(p->np + 1))) != WXA_OK) return rc;
        // the histogram and, behind it, the foreign counters; the own counts per cell
        if ((rc = s.hist.reserve(sizeof(int) * 2 * (ncells + 2))) != WXA_OK) return rc;
        if ((rc = s.offs[s.out].reserve(sizeof(int) * (ncells + 2))) != WXA_OK) return rc;
        if ((rc = s.own[s.out].reserve(sizeof(int) * (ncells + 2))) != WXA_OK) return rc;
        WXA_HIP_CHECK(hipMemsetAsync(s.hist.p, 0, sizeof(int) * 2 * (ncells + 2), st));
        WXA_HIP_CHECK(hipMemsetAsync(s.own[s.out].p, 0, sizeof(int) * (ncells + 2), st));
        s.check_retired = check_retired && p->idcpu ? 1 : 0;
        s.predict_dt = predict_dt;
        for (int d = 0; d < 3; ++d) {
            s.plo[d] = plo[d]; s.dinv[d] = dinv[d]; s.nc[d] = ncell[d]; s.cell_lo[d] = cell_lo[d]; s.wrap[d] = wrap[d] ? 1 : 0;
        }
        s.bins = ncells;
    }
    s.np_armed = p->np;
    s.count_x = p->x;
    s.armed = mode;
    return WXA_OK;
}

wxa_status wxa_push_sort_end(wxa_workspace* ws, int32_t read_live, int64_t* live, int64_t* appended, void* stream) {
    WXA_REQUIRE(ws && live && appended, "null argument");
    auto& s = ws->ps;
    WXA_REQUIRE(s.armed != 0, "wxa_push_sort_end without wxa_push_sort_begin");
    hipStream_t st = (hipStream_t)stream;
    const int32_t mode = s.armed;
    s.armed = 0;
    *live = s.np_armed;
    *appended = 0;
    const double* tile_x = nullptr;   // identity of the tile a new record indexes
    int64_t tile_np = s.np_armed;
    if (mode & WXA_PUSH_SORT_SCATTER) {
        // the record's scan becomes the tile offsets of the LDS-tile kernels; the workspace now describes the destination tile
        std::swap(ws->offsets.p, s.offs[s.in].p);
        std::swap(ws->offsets.cap, s.offs[s.in].cap);
        int64_t n_live = s.pending_np;
        if (read_live) {   // retired particles were counted: the cell-sorted part ends where their bin starts
            int v = 0;
            WXA_HIP_CHECK(hipMemcpyAsync(&v, (const int*)ws->offsets.p + s.pending_bins, sizeof(int), hipMemcpyDeviceToHost, st));
            WXA_HIP_CHECK(hipStreamSynchronize(st));
            n_live = v;
        }
        ws->sorted_valid = true;
        ws->sorted_np = n_live;
        ws->sorted_bins = s.pending_bins;
        ws->sorted_x = s.dst.x;
        for (int d = 0; d < 3; ++d) {
            ws->sort_nc[d] = s.p_nc[d]; ws->sort_cell_lo[d] = s.p_cell_lo[d];
            ws->sort_plo[d] = s.p_plo[d]; ws->sort_dinv[d] = s.p_dinv[d];
        }
        s.pending = false;
        *live = n_live;
        *appended = s.appended;
        tile_x = s.dst.x;
        tile_np = n_live + s.appended;
    }
    if (mode & WXA_PUSH_SORT_COUNT) {
        size_t tmp_bytes = 0;
        int* hist = (int*)s.hist.p;
        int* offs = (int*)s.offs[s.out].p;
        WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, hist, offs, (int)(s.bins + 2), st));
        wxa_status rc;
        if ((rc = ws->scan_tmp.reserve(tmp_bytes)) != WXA_OK) return rc;
        WXA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ws->scan_tmp.p, tmp_bytes, hist, offs, (int)(s.bins + 2), st));
        s.in = s.out;
        s.pending = true;
        s.pending_np = tile_np;
        s.pending_bins = s.bins;
        s.pending_x = tile_x ? tile_x : nullptr;   // COUNT alone: set by the caller's view, below
        for (int d = 0; d < 3; ++d) {
            s.p_nc[d] = s.nc[d]; s.p_cell_lo[d] = s.cell_lo[d]; s.p_plo[d] = s.plo[d]; s.p_dinv[d] = s.dinv[d];
        }
        if (!tile_x) s.pending_x = s.count_x;
    }
    return WXA_OK;
}

int32_t wxa_push_sort_pending(const wxa_workspace* ws, const wxa_particle_view* p) {
    return ws && p && ws->ps.pending && ws->ps.pending_x == p->x && ws->ps.pending_np <= p->np ? 1 : 0;
}

wxa_status wxa_sort_live_count(wxa_workspace* ws, int64_t* n, void* stream) {
    WXA_REQUIRE(ws && n, "null argument");
    WXA_REQUIRE(ws->sorted_valid, "no sort recorded in this workspace");
    int live = 0;
    hipStream_t st = (hipStream_t)stream;
    WXA_HIP_CHECK(hipMemcpyAsync(&live, (const int*)ws->offsets.p + ws->sorted_bins, sizeof(int),
                                 hipMemcpyDeviceToHost, st));
    WXA_HIP_CHECK(hipStreamSynchronize(st));
    ws->sorted_np = live;
    *n = live;
    return WXA_OK;
}

}  // extern "C"
