// LDS-tile gather + push, two particles per lane (gfx950).
//
// The tile kernel of gather_tile.hip is bound by LDS read issue: 252 field values per particle at order 3 with the
// energy-conserving gather = 126 ds_read2_b64 per lane, 3.4 ms for 1.3e8 particles at the LDS's 128 B/clk even when
// nothing else stalls.  With an odd shape order O and the Galerkin gather (cell-centred weights of order O - 1) every
// stencil frame of a particle is a function of its CELL alone -- the nodal weights start at floor(x) - (O-1)/2, the
// cell-centred ones at round(x - 1/2) - (O-1)/2 = floor(x) - (O-1)/2 -- so the particles of one cell of the sort read
// exactly the same field values.  Here a lane takes two particles of one cell, reads every stencil row ONCE and
// accumulates it into both particles' sums with their own weights (the one-particle kernel's arithmetic up to the
// association of the two transverse weights): half the LDS instructions for the same arithmetic.
//
// Lane mapping as in deposit_tile_rows_kernel: chunk b = 16 consecutive cells x their first four pairs, lane (r, c) takes
// the particles (2 r, 2 r + 1) of cell 16 b + c (loads cover the block contiguously); pairs beyond a cell's fourth come
// from a tail table built from the cell counts (one ballot per row, one wave scan).  A second particle that has left
// its partner's cell since the sort goes to a small list and is gathered alone after the loop; stencils that leave
// the staged tile go to the straggler queue of gather_tile.hip (global loads).
// Reference: doGatherShapeN (Source/Particles/Gather/FieldGather.H:36-424), PushPX / PushP
// (Source/Particles/PhysicalParticleContainer.cpp:2549-2786, 2368-2516).
#include "gather_body.hpp"
#include "workspace.hpp"

#include <stdlib.h>

namespace wxa {

constexpr int GP_TS = WXA_TILE;
constexpr int GP_THREADS = 384;   // 6 waves; 2 workgroups per CU (2 x 64 KB LDS), 3 waves per SIMD (<= 168 VGPRs)
constexpr int GP_CELLS = GP_TS * GP_TS * GP_TS;

struct GPTileGeom {
    int nt[3];
    int cell_lo[3];
};
struct GPStragglers {
    int* __restrict__ idx;
    unsigned* __restrict__ count;
    __device__ __forceinline__ void push(int ip) const { idx[atomicAdd(count, 1u)] = ip; }
};

// sum_{iz,iy,ix} s?x[ix] s?y[iy] s?z[iz] F(ix,iy,iz) for two particles that share the block: every row is read once
template <int NX, int NY, int NZ>
__device__ __forceinline__ void gather_rows2(const double* __restrict__ base, long js, long ks,
                                             const double* __restrict__ ax, const double* __restrict__ ay,
                                             const double* __restrict__ az, const double* __restrict__ bx,
                                             const double* __restrict__ by, const double* __restrict__ bz,
                                             double& fa, double& fb) {
    double acca = 0.0, accb = 0.0;
#pragma unroll
    for (int iz = 0; iz < NZ; ++iz) {
#pragma unroll
        for (int iy = 0; iy < NY; ++iy) {
            const double* __restrict__ row = base + iy * js + iz * ks;
            double v[NX];
#pragma unroll
            for (int ix = 0; ix < NX; ++ix) v[ix] = row[ix];
            double ra = 0.0, rb = 0.0;
#pragma unroll
            for (int ix = 0; ix < NX; ++ix) { ra += ax[ix] * v[ix]; rb += bx[ix] * v[ix]; }
            // sy (sz r), not (sy sz) r: the products sy sz are common to several components, and the compiler keeps all
            // 49 of them per particle alive across the six gathers when they can be shared (196 VGPRs for a pair)
            acca += ay[iy] * (az[iz] * ra);
            accb += by[iy] * (bz[iz] * rb);
        }
        // keep the scheduler from hoisting the next planes' reads above this plane's arithmetic: with everything in
        // flight the 252 values of a particle pair want 500 VGPRs (measured: 435 spilled at the 168 budget)
        __builtin_amdgcn_sched_barrier(0);
    }
    fa = acca; fb = accb;
}

template <int PUSHER, bool MOVE>
__device__ __forceinline__ void gp_push_store(const PV& p, int ip, double xp, double yp, double zp, double Exp, double Eyp,
                                              double Ezp, double Bxp, double Byp, double Bzp, double q, double m,
                                              double dt) {
    double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip];
    push_momentum<PUSHER>(ux, uy, uz, Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, m, dt);
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz;
    if constexpr (MOVE) {
        update_position(xp, yp, zp, ux, uy, uz, dt);
        p.x[ip] = xp; p.y[ip] = yp; p.z[ip] = zp;
    }
}

template <int O, int PUSHER, bool MOVE>
__global__ void __launch_bounds__(GP_THREADS) WXA_WAVES_PER_SIMD(3)
gather_push_pairs_kernel(PV p, const int* __restrict__ offsets, DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz,
                         Geom g, GPTileGeom tg, double q, double m, double dt, GPStragglers sq, ExtEB ext) {
    static_assert(O == 1 || O == 3, "cell-uniform stencil frames need an odd order (and the Galerkin gather)");
    constexpr int G = 1;
    constexpr int N = GP_TS + 3;                   // staged points per direction (GatherTileDims<1>)
    constexpr int LO = -1;
    constexpr int NPTS = N * N * N;
    constexpr int NN = O + 1, NC = O;              // nodal / cell-centred weights per direction
    constexpr int WAVES = GP_THREADS / 64;
    constexpr int CW = GP_CELLS / 64;              // cell-waves of the table phases
    constexpr int RT = 64 / CW;                    // tail rows (pairs 4 .. 3 + RT)
    constexpr int NB = GP_CELLS / 16;
    constexpr int TCAP = GP_CELLS * 2;
    constexpr int LCAP = 1024;
    __shared__ double F[6 * NPTS];
    __shared__ unsigned long long masks[RT][CW];
    __shared__ int cstart[GP_CELLS + 1];
    __shared__ unsigned short table[TCAP];
    __shared__ unsigned lone[LCAP];
    __shared__ int nlone, nitems;
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const long tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const long ucell0 = tile * GP_CELLS;
    const int start = offsets[ucell0];
    const int end = offsets[ucell0 + GP_CELLS];
    if (end <= start) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ti = (int)(tile % tg.nt[0]);
    const int tj = (int)((tile / tg.nt[0]) % tg.nt[1]);
    const int tk = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
    const int o0 = tg.cell_lo[0] + ti * GP_TS + LO;
    const int o1 = tg.cell_lo[1] + tj * GP_TS + LO;
    const int o2 = tg.cell_lo[2] + tk * GP_TS + LO;
    auto put_lone = [&](const int ip) {
        const int n = atomicAdd(&nlone, 1);
        if (n < LCAP) lone[n] = (unsigned)ip;
        else sq.push(ip);
    };
    // ---- cell counts and the tail rows (one lane per cell: 512 cells over 384 lanes in two rounds), field staging
    if (tid == 0) { nlone = 0; nitems = 0; }
    // phase A runs on the first CW waves' worth of cells per round
    for (int c0 = 0; c0 < GP_CELLS; c0 += GP_THREADS) {
        const int c = c0 + tid;
        if (c < GP_CELLS) {
            const int s = offsets[ucell0 + c];
            cstart[c] = s;
            if (c == GP_CELLS - 1) cstart[GP_CELLS] = offsets[ucell0 + GP_CELLS];
        }
    }
    const DevF* fld[6] = {&Ex, &Ey, &Ez, &Bx, &By, &Bz};
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const DevF& f = *fld[c];
        for (int a = tid; a < NPTS; a += GP_THREADS) {
            const int i = o0 + a % N, j = o1 + (a / N) % N, k = o2 + a / (N * N);
            const bool in = i >= f.lo0 && i < f.lo0 + f.n0 && j >= f.lo1 && j < f.lo1 + f.n1 && k >= f.lo2 &&
                            k < f.lo2 + f.n2;
            F[c * NPTS + a] = in ? f.p[f.off(i, j, k)] : 0.0;
        }
    }
    __syncthreads();
    // row masks: cell-wave w = cells 64 w .. 64 w + 63, handled by wave w % WAVES in round w / WAVES
    for (int cw = wave; cw < CW; cw += WAVES) {
        const int c = 64 * cw + lane;
        const int n = cstart[c + 1] - cstart[c];
        const int pairs = min((n + 1) >> 1, 4 + RT);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const unsigned long long mk = __ballot(pairs > 4 + r);
            if (lane == 0) masks[r][cw] = mk;
        }
        for (int k = 2 * (4 + RT); k < n; ++k) put_lone(cstart[c] + k);   // beyond the table's rows
    }
    __syncthreads();
    for (int cw = wave; cw < CW; cw += WAVES) {
        const int cnt = __popcll(masks[lane / CW][lane % CW]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int excl = incl - cnt;
        if (cw == 0 && lane == 63) nitems = min(incl, TCAP);
        const int c = 64 * cw + lane;
        const int s = cstart[c], n = cstart[c + 1] - s;
        const int pairs = min((n + 1) >> 1, 4 + RT);
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int base = __shfl(excl, r * CW + cw);
            if (pairs > 4 + r) {
                const int at = base + __popcll(masks[r][cw] & lt);
                if (at < TCAP) table[at] = (unsigned short)(c | ((4 + r) << 9));
                else { put_lone(s + 2 * (4 + r)); if (2 * (4 + r) + 1 < n) put_lone(s + 2 * (4 + r) + 1); }
            }
        }
    }
    __syncthreads();
    // ---- the chunks: mode 0 = blocks and tail (pairs), mode 1 = the lone particles
    const int T = nitems;
    const int nchunks = NB + ((T + 63) >> 6);
    int mode = 0, ch = wave, nl = 0;
    for (;;) {
        int ia = start;
        bool va = false, vb = false;
        if (mode == 0) {
            if (ch >= nchunks) {   // every wave passes here exactly once
                __syncthreads();
                mode = 1;
                ch = wave;
                nl = min(nlone, LCAP);
                continue;
            }
            int c, r;
            if (ch < NB) {
                c = 16 * ch + (lane & 15); r = lane >> 4; va = true;
            } else {
                const int I = (ch - NB) * 64 + lane;
                va = I < T;
                const unsigned ent = va ? table[I] : 0u;
                c = (int)(ent & 511u); r = (int)(ent >> 9);
            }
            const int s0 = cstart[c], n0 = cstart[c + 1] - s0;
            va = va && 2 * r < n0;
            vb = va && 2 * r + 1 < n0;
            if (va) ia = s0 + 2 * r;
        } else {
            if (ch * 64 >= nl) break;
            const int it = ch * 64 + lane;
            if (it < nl) { ia = (int)lone[it]; va = true; }
        }
        const int ib = vb ? ia + 1 : ia;
        const double xa = p.x[ia], ya = p.y[ia], za = p.z[ia];
        const double xb = p.x[ib], yb = p.y[ib], zb = p.z[ib];
        GatherShapes<O, G> sa, sb;
        gather_shapes<O, G>(xa, ya, za, g, sa);
        gather_shapes<O, G>(xb, yb, zb, g, sb);
        auto staged = [&](const GatherShapes<O, G>& s) {
            const int lo_i = min(s.jn, s.jc) - o0, hi_i = max(s.jn + NN, s.jc + NC) - 1 - o0;
            const int lo_j = min(s.kn, s.kc) - o1, hi_j = max(s.kn + NN, s.kc + NC) - 1 - o1;
            const int lo_k = min(s.ln, s.lc) - o2, hi_k = max(s.ln + NN, s.lc + NC) - 1 - o2;
            return lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < N && hi_j < N && hi_k < N;
        };
        // the second particle shares the reads only on its partner's frame; elsewhere it is gathered alone later
        const bool same = sa.jn == sb.jn && sa.jc == sb.jc && sa.kn == sb.kn && sa.kc == sb.kc && sa.ln == sb.ln &&
                          sa.lc == sb.lc;
        if (vb && !same) { if (mode == 0) put_lone(ib); vb = false; }   // (mode 1 has no second particle)
        if (va && !staged(sa)) {
            sq.push(ia);
            if (vb) { if (mode == 0) put_lone(ib); vb = false; }
            va = false;
        }
        if (va) {
            const int jn = sa.jn - o0, jc = sa.jc - o0, kn = sa.kn - o1, kc = sa.kc - o1, ln = sa.ln - o2, lc = sa.lc - o2;
            double Exa, Eya, Eza, Bxa, Bya, Bza, Exb, Eyb, Ezb, Bxb, Byb, Bzb;
            gather_rows2<NC, NN, NN>(F + 0 * NPTS + jc + N * (kn + N * ln), N, N * N, sa.sxc, sa.syn, sa.szn, sb.sxc, sb.syn, sb.szn, Exa, Exb);
            gather_rows2<NN, NC, NN>(F + 1 * NPTS + jn + N * (kc + N * ln), N, N * N, sa.sxn, sa.syc, sa.szn, sb.sxn, sb.syc, sb.szn, Eya, Eyb);
            gather_rows2<NN, NN, NC>(F + 2 * NPTS + jn + N * (kn + N * lc), N, N * N, sa.sxn, sa.syn, sa.szc, sb.sxn, sb.syn, sb.szc, Eza, Ezb);
            gather_rows2<NC, NC, NN>(F + 5 * NPTS + jc + N * (kc + N * ln), N, N * N, sa.sxc, sa.syc, sa.szn, sb.sxc, sb.syc, sb.szn, Bza, Bzb);
            gather_rows2<NC, NN, NC>(F + 4 * NPTS + jc + N * (kn + N * lc), N, N * N, sa.sxc, sa.syn, sa.szc, sb.sxc, sb.syn, sb.szc, Bya, Byb);
            gather_rows2<NN, NC, NC>(F + 3 * NPTS + jn + N * (kc + N * lc), N, N * N, sa.sxn, sa.syc, sa.szc, sb.sxn, sb.syc, sb.szc, Bxa, Bxb);
            // The second particle's sums are used only under `if (vb)`: left alone, the compiler sinks their whole
            // arithmetic into that branch -- far behind the reads it shares with the first particle, which then have to
            // survive in scratch (435 spilled VGPRs).  Pin the sums where they are computed.
            WXA_OPAQUE_F64(Exb); WXA_OPAQUE_F64(Eyb); WXA_OPAQUE_F64(Ezb);
            WXA_OPAQUE_F64(Bxb); WXA_OPAQUE_F64(Byb); WXA_OPAQUE_F64(Bzb);
            gp_push_store<PUSHER, MOVE>(p, ia, xa, ya, za, Exa + ext.ex, Eya + ext.ey, Eza + ext.ez, Bxa + ext.bx, Bya + ext.by,
                                        Bza + ext.bz, q, m, dt);
            if (vb)
                gp_push_store<PUSHER, MOVE>(p, ib, xb, yb, zb, Exb + ext.ex, Eyb + ext.ey, Ezb + ext.ez, Bxb + ext.bx,
                                            Byb + ext.by, Bzb + ext.bz, q, m, dt);
        }
        ch += WAVES;
    }
}

template <int O, int PUSHER, bool MOVE>
static wxa_status launch_pairs(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                               const wxa_grid_geom* geom, double q, double m, double dt, wxa_workspace* ws, hipStream_t st) {
    GPTileGeom tg;
    for (int d = 0; d < 3; ++d) {
        tg.nt[d] = (ws->sort_nc[d] + GP_TS - 1) / GP_TS;
        tg.cell_lo[d] = ws->sort_cell_lo[d];
    }
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const PV pv = make_pv(*p);
    const Geom g = make_geom(*geom);
    const int* offsets = (const int*)ws->offsets.p;
    const DevF ex = make_devf(E[0]), ey = make_devf(E[1]), ez = make_devf(E[2]);
    const DevF bx = make_devf(B[0]), by = make_devf(B[1]), bz = make_devf(B[2]);
    const dim3 grid((unsigned)xcd_grid_size(ntiles)), block(GP_THREADS);
    // the straggler queue (reserved and reset by the caller, gather_tile.hip, which also runs the straggler pass)
    GPStragglers sq{(int*)ws->stragglers.p, (unsigned*)ws->counters.p + 16};
    hipLaunchKernelGGL((gather_push_pairs_kernel<O, PUSHER, MOVE>), grid, block, 0, st, pv, offsets, ex, ey, ez, bx, by, bz,
                       g, tg, q, m, dt, sq, ext_of(ws));
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// 1 = handled here (the caller still runs the straggler pass of gather_tile.hip on the queue), 0 = not this configuration
bool gather_pairs_applicable(int order, int galerkin, int pusher) {
    if (getenv("WXA_GATHER_PAIRS") && atoi(getenv("WXA_GATHER_PAIRS")) == 0) return false;
    return galerkin == 1 && (order == 1 || order == 3) && (pusher == WXA_PUSHER_BORIS || pusher == WXA_PUSHER_VAY);
}

wxa_status gather_push_pairs(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                             const wxa_grid_geom* geom, double q, double m, double dt, int order, int pusher, bool move,
                             wxa_workspace* ws, hipStream_t st) {
#define WXA_GP(O, PU)                                                                        \
    (move ? launch_pairs<O, PU, true>(p, E, B, geom, q, m, dt, ws, st)                        \
          : launch_pairs<O, PU, false>(p, E, B, geom, q, m, dt, ws, st))
    if (pusher == WXA_PUSHER_BORIS) return order == 1 ? WXA_GP(1, WXA_PUSHER_BORIS) : WXA_GP(3, WXA_PUSHER_BORIS);
    return order == 1 ? WXA_GP(1, WXA_PUSHER_VAY) : WXA_GP(3, WXA_PUSHER_VAY);
#undef WXA_GP
}

}  // namespace wxa
