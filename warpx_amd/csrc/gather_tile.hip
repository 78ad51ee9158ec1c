// LDS-tile gather + push for gfx950.
//
// The plain gather (particles.hip) issues 252 global loads per particle at order 3; with
// cell-sorted particles they hit L1, but the texture-address path moves 64 lanes x 8 B per
// instruction at ~40 B/clk, which bounds the kernel (rocprof: ~13 clk per wave load, VALU 25 %
// busy).  Here one workgroup owns one 8x8x8-cell tile of the tile-major cell sort, stages the six
// staggered field components of the tile plus the stencil halo in LDS once
// (6 x 11^3 x 8 B = 63.9 KB with the energy-conserving gather, 2 workgroups per CU) and serves the
// gathers from LDS (same-cell lanes broadcast).  Particles whose stencil leaves the staged range
// (stale sort, particles wrapped across the periodic boundary) are queued and handled by a second
// kernel with global loads, so correctness never depends on the sort being fresh.
// The rows are read with single ds_read_b64 on a fixed software pipeline (gather_rows_lds, gather_body.hpp; round 3).
// (Tried and rejected, ms per launch at 256^3 x 8 ppc against 5.9: round 2's first attempt at single reads -- inline asm
// with one s_waitcnt lgkmcnt(0) per row and nothing in flight behind it -- 6.3; 640 lanes per tile = 5 waves per SIMD, 5.9;
// the next particle's position and momentum loaded while the current one gathers (127 VGPRs), 6.2; this particle's
// momentum loaded together with its position instead of after the gather (hand-issued loads), no change; two particles of a
// cell per lane sharing the LDS reads, 9.95.  Counters, profiles/round2/r2g_pmc_128cube_gather_tile_kernel.txt: 8 LDS
// cycles per ds instruction and 0.2 % bank conflicts, the LDS busy 60 % and the VALU 50 % of the kernel's time.)
#include "gather_body.hpp"
#include "heavy_tiles.hpp"
#include "push_sort.hpp"
#include "workspace.hpp"

#include <stdlib.h>

#ifndef WXA_STRAGGLER_BLOCKS
// workgroups of 256 lanes of gather_push_stragglers_kernel (a grid-stride loop over a list whose length only the device
// knows).  512 workgroups are two waves per SIMD; 2048 were measured and change nothing (0.19-0.62 ms per launch either
// way at 256^3 x 8 per cell, profiles/round5/README.md): the kernel is not short of waves in flight
#define WXA_STRAGGLER_BLOCKS 512
#endif
namespace wxa {

// Clocks of the tile kernel (opt-in build, -DWXA_GATHER_PROFILE; read with scripts/gather_profile.py): thread 0 of every
// workgroup adds [0] start -> field tile staged, [1] staged -> wave 0 out of its particle loop, [2] workgroups; per trip of
// wave 0: [3] its particle's loads, [4] shapes + LDS gather, [5] momentum loads + push + stores, [6] trips
#ifdef WXA_GATHER_PROFILE
__device__ unsigned long long wxa_gather_prof[8];
#define GPROF_CLOCK(v) const long long v = clock64()
#define GPROF_ADD(n, v) do { if (threadIdx.x == 0) atomicAdd(&wxa_gather_prof[n], (unsigned long long)(v)); } while (0)
#else
#define GPROF_CLOCK(v)
#define GPROF_ADD(n, v)
#endif

constexpr int GT_TS = WXA_TILE;
constexpr int GT_THREADS = 512;

// Staged points per direction: the reach of the stencils of the particles of the tile's own cells, nothing more.
// Orders 1 - 3 (the node below for odd orders; the widest is order 3): nodes c - 1 .. c + 2 of a particle's cell c, and
// with the same shape at the staggered points (G = 0) c - 2 .. c + 2.  Order 4 (round 6; the nearest node): nodes
// c - 2 .. c + 3, staggered points c - 2 .. c + 2 with the Galerkin gather's cubic shape there, c - 3 .. c + 2 without.
// LDS per workgroup: 64 / 83 KB (orders <= 3: two workgroups per CU with G = 1), 105 / 132 KB (order 4: one).
template <int G, int O = 3>
struct GatherTileDims {
    static constexpr int LO = O == 4 ? (G ? -2 : -3) : (G ? -1 : -2);   // first staged point relative to the tile's first cell
    static constexpr int N = GT_TS + (O == 4 ? (G ? 5 : 6) : (G ? 3 : 4));   // staged points per direction
    static constexpr int NPTS = N * N * N;
};

// Orders <= 3 without the Galerkin shapes (G = 0: what the reference pairs with direct deposition): a component needs the
// twelve points -2 .. 9 only along the directions in which it is cell-centred, eleven (-1 .. 9) along the others.  With the
// arrays sized per component the stage is 3 x 12 x 11 x 11 + 3 x 11 x 12 x 12 doubles = 73 KB instead of 6 x 12^3 = 83 KB:
// two workgroups per CU again, like the Galerkin gather (round 6; the direct-deposition line's gather 7.6 ms before).
struct SplitTile {
    static constexpr int stag(int c, int d) {   // Ex Ey Ez Bx By Bz: cell-centred along d?
        return c == 0 ? d == 0 : c == 1 ? d == 1 : c == 2 ? d == 2 : c == 3 ? d != 0 : c == 4 ? d != 1 : d != 2;
    }
    static constexpr int n(int c, int d) { return stag(c, d) ? GT_TS + 4 : GT_TS + 3; }
    static constexpr int lo(int c, int d) { return stag(c, d) ? -2 : -1; }
    static constexpr int pts(int c) { return n(c, 0) * n(c, 1) * n(c, 2); }
    static constexpr int off(int c) { return c == 0 ? 0 : off(c - 1) + pts(c - 1); }
    static constexpr int total() { return off(6); }
};

struct GTileGeom {
    int nt[3];
    int cell_lo[3];
};

template <int N>
struct LdsField {
    const double* base;
    __device__ __forceinline__ const double* at(int i, int j, int k) const { return base + i + N * (j + N * k); }
    static constexpr long js = N, ks = N * N;
};

struct GatherStragglers {
    int* __restrict__ idx;
    unsigned* __restrict__ count;
    unsigned* __restrict__ next = nullptr;   // the next launch's counter, zeroed by this one (wxa::flip_counter)
    __device__ __forceinline__ void push(int ip) const { idx[atomicAdd(count, 1u)] = ip; }
};

template <int PUSHER, bool MOVE>
__device__ __forceinline__ void push_and_store(const PV& p, int ip, double xp, double yp, double zp, double ux, double uy,
                                               double uz, double Exp, double Eyp, double Ezp, double Bxp, double Byp,
                                               double Bzp, double q, double m, double dt, const ExtEB& ext,
                                               const PushSort& hook,
                                               int* lds_hist = nullptr, const long my_tile = -1) {   // push_sort_tail
    add_external_fields(ext, ip, Exp, Eyp, Ezp, Bxp, Byp, Bzp);
    push_momentum<PUSHER>(ux, uy, uz, Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, m, dt);
    if constexpr (MOVE) update_position(xp, yp, zp, ux, uy, uz, dt);
    if constexpr (MOVE) {   // the cell sort folded into the push (push_sort.hpp): keyed, or written to the sorted tile
        if (!push_sort_tail(hook, p, ip, xp, yp, zp, ux, uy, uz, lds_hist, my_tile)) return;
    }
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz;
    if constexpr (MOVE) { p.x[ip] = xp; p.y[ip] = yp; p.z[ip] = zp; }
}

// 512 threads per tile and no global-load fallback inside (127 VGPRs at order 3): two workgroups
// per CU = 4 waves per SIMD, which is what hides the LDS read latency of the 252-point gather.
// PART: 0 = every tile; 1 = only the tiles that touch no face of the sorted box (their particles read no guard
// point, even as stragglers: a particle is at most a few cells from the tile it was sorted into); 2 = only
// the tiles that do.  1 and 2 let the guard exchange of E and B travel behind the interior tiles
// (wxa_gather_push_part); the default path instantiates PART = 0 and is unchanged by them.
template <int O, int G, int PUSHER, bool MOVE, int PART = 0>
__global__ void __launch_bounds__(GT_THREADS) WXA_WAVES_PER_SIMD(O <= 3 ? 4 : 2)   // what the staged tile lets a CU hold
gather_push_tile_kernel(PV p, const int* __restrict__ offsets, DevF Ex, DevF Ey, DevF Ez, DevF Bx, DevF By,
                        DevF Bz, Geom g, GTileGeom tg, double q, double m, double dt, GatherStragglers sq, ExtEB ext,
                        PushSort hook, HeavyUnits hu) {
    constexpr int N = GatherTileDims<G, O>::N;
    constexpr int NPTS = GatherTileDims<G, O>::NPTS;
    constexpr bool SPLIT = G == 0 && O <= 3;   // per-component stage (SplitTile)
    __shared__ double F[SPLIT ? SplitTile::total() : 6 * NPTS];
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    if (blockIdx.x == 0 && threadIdx.x == 0 && sq.next) *sq.next = 0u;
    // a tile with far more particles than the others is shared by several workgroups, each with a part of its particles
    // (heavy_tiles.hpp)
    long tile;
    int unit_u, unit_k;
    if (!heavy_unit_of(hu, blockIdx.x, ntiles, tile, unit_u, unit_k)) return;
    constexpr int TC = GT_TS * GT_TS * GT_TS;
    int start = offsets[tile * TC];
    int end = offsets[(tile + 1) * TC];
    if (unit_k > 1) {   // parts of whole 64-particle chunks
        const long chunks = ((long)(end - start) + 63) >> 6;
        const int s0 = start;
        start = s0 + (int)((chunks * unit_u / unit_k) << 6);
        end = min(end, s0 + (int)((chunks * (unit_u + 1) / unit_k) << 6));
    }
    if (end <= start) return;
    const int tid = threadIdx.x;
    GPROF_CLOCK(prof_t0);
    const int ti = (int)(tile % tg.nt[0]);
    const int tj = (int)((tile / tg.nt[0]) % tg.nt[1]);
    const int tk = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
    if constexpr (PART != 0) {
        const bool face = ti == 0 || ti == tg.nt[0] - 1 || tj == 0 || tj == tg.nt[1] - 1 || tk == 0 || tk == tg.nt[2] - 1;
        if (face != (PART == 2)) return;
    }
    const int o0 = tg.cell_lo[0] + ti * GT_TS + GatherTileDims<G, O>::LO;
    const int o1 = tg.cell_lo[1] + tj * GT_TS + GatherTileDims<G, O>::LO;
    const int o2 = tg.cell_lo[2] + tk * GT_TS + GatherTileDims<G, O>::LO;
    // Staging: all loads of a component pair are in flight before the first LDS write (as a plain
    // `for (a = tid; ...) F[a] = load` loop every lane had one load in flight at a time).
    // The lane's particle of the NEXT trip is requested while the current one gathers (plain loads between compiler
    // fences, so that they keep their place in front of the inline-asm LDS reads), the first one before the staging: no
    // trip starts by waiting for HBM (without it the waves were parked in s_waitcnt 46 % of their cycles); this trip's
    // momentum is requested at the top of the trip and used after the gather.
    // The 64-particle chunks of the tile are handed out through an LDS counter instead of chunk = wave + 8 k (the SIMD's
    // arbiter favours its oldest waves; a workgroup's LDS is held until its slowest wave is done): a wave works on one
    // chunk, has the positions of the next one in flight and has claimed the one after that -- the counter's round trip
    // through the LDS queue hides behind a whole trip.  The first two chunks of every wave are the static ones.
    // A tile's stragglers are collected in LDS and written to the global list as ONE contiguous block at the end of the
    // workgroup (one global atomic per tile): the straggler kernel's waves then hold particles of one or two neighbouring
    // tiles, whose 252 scattered loads share cache lines.
    // The sort folded into the push, COUNT: ranks from a histogram of the tile's own cells (push_sort.hpp).
    static_assert(GT_THREADS == PUSH_SORT_TILE_CELLS, "one lane per cell of the tile");
    __shared__ int lhist[MOVE ? PUSH_SORT_TILE_CELLS : 1];
    // (COUNT alone or together with SCATTER: the keys are those of the tile being read either way)
    const bool count_local = MOVE && (hook.mode & PUSH_SORT_COUNT) != 0 && unit_u == 0;   // uniform (a shared tile: unit 0's histogram)
    if constexpr (MOVE) {
        if (count_local) lhist[tid] = 0;   // visible after the staging barrier below
    }
    constexpr int SCAP = 512;
    __shared__ int slist[SCAP];
    __shared__ int sn, sbase;
    if (tid == 0) sn = 0;   // visible after the staging barrier below
    auto push_straggler = [&](const int i) {
        const int n = atomicAdd(&sn, 1);
        if (n < SCAP) slist[n] = i;
        else sq.push(i);
    };
    constexpr int WAVES = GT_THREADS / 64;
    __shared__ int next_chunk;
    const int lane = tid & 63;
    int cb_next = start + 64 * __builtin_amdgcn_readfirstlane(tid >> 6) + GT_THREADS;   // wave-uniform: first particle of the next chunk
    if (tid == 0) next_chunk = 2 * WAVES;   // visible after the staging barrier below
    int ip = start + tid;
    double nxt[3] = {0.0, 0.0, 0.0};
    // plain loads between two compiler fences: they stay where they are written (in front of the inline-asm LDS reads) and
    // keep the L1/L2 path of ordinary loads (volatile loads become flat_load ... sc0 sc1: system-coherent, slower)
    auto load_particle = [&](const int i) {
        asm volatile("" ::: "memory");
        nxt[0] = p.x[i]; nxt[1] = p.y[i]; nxt[2] = p.z[i];
        asm volatile("" ::: "memory");
    };
    if (ip < end) load_particle(ip);
    constexpr int PER = (NPTS + GT_THREADS - 1) / GT_THREADS;
    auto fetch = [&](const DevF& f, double (&r)[PER]) {
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int a = tid + n * GT_THREADS;
            const int i = o0 + a % N, j = o1 + (a / N) % N, k = o2 + a / (N * N);
            const bool in = a < NPTS && i >= f.lo0 && i < f.lo0 + f.n0 && j >= f.lo1 && j < f.lo1 + f.n1 && k >= f.lo2 &&
                            k < f.lo2 + f.n2;
            r[n] = in ? f.p[f.off(i, j, k)] : 0.0;
        }
    };
    auto put = [&](int c, const double (&r)[PER]) {
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const int a = tid + n * GT_THREADS;
            if (a < NPTS) F[c * NPTS + a] = r[n];
        }
    };
    if constexpr (SPLIT) {
        const int t0 = o0 - GatherTileDims<G, O>::LO, t1 = o1 - GatherTileDims<G, O>::LO, t2 = o2 - GatherTileDims<G, O>::LO;   // the tile's first cell
        constexpr int PERS = (SplitTile::pts(3) + GT_THREADS - 1) / GT_THREADS;
        auto fetch_c = [&](auto cc, const DevF& f, double (&r)[PERS]) {
            constexpr int c = decltype(cc)::value, n0 = SplitTile::n(c, 0), n1 = SplitTile::n(c, 1);
#pragma unroll
            for (int n = 0; n < PERS; ++n) {
                const int a = tid + n * GT_THREADS;
                const int i = t0 + SplitTile::lo(c, 0) + a % n0, j = t1 + SplitTile::lo(c, 1) + (a / n0) % n1,
                          k = t2 + SplitTile::lo(c, 2) + a / (n0 * n1);
                const bool in = a < SplitTile::pts(c) && i >= f.lo0 && i < f.lo0 + f.n0 && j >= f.lo1 && j < f.lo1 + f.n1 &&
                                k >= f.lo2 && k < f.lo2 + f.n2;
                r[n] = in ? f.p[f.off(i, j, k)] : 0.0;
            }
        };
        auto put_c = [&](auto cc, const double (&r)[PERS]) {
            constexpr int c = decltype(cc)::value;
#pragma unroll
            for (int n = 0; n < PERS; ++n) {
                const int a = tid + n * GT_THREADS;
                if (a < SplitTile::pts(c)) F[SplitTile::off(c) + a] = r[n];
            }
        };
        using std::integral_constant;
        double r0[PERS], r1[PERS], r2[PERS], r3[PERS], r4[PERS], r5[PERS];
        fetch_c(integral_constant<int, 0>{}, Ex, r0); fetch_c(integral_constant<int, 1>{}, Ey, r1); fetch_c(integral_constant<int, 2>{}, Ez, r2);
        fetch_c(integral_constant<int, 3>{}, Bx, r3); fetch_c(integral_constant<int, 4>{}, By, r4); fetch_c(integral_constant<int, 5>{}, Bz, r5);
        put_c(integral_constant<int, 0>{}, r0); put_c(integral_constant<int, 1>{}, r1); put_c(integral_constant<int, 2>{}, r2);
        put_c(integral_constant<int, 3>{}, r3); put_c(integral_constant<int, 4>{}, r4); put_c(integral_constant<int, 5>{}, r5);
    } else {
        double r0[PER], r1[PER], r2[PER], r3[PER], r4[PER], r5[PER];
        fetch(Ex, r0); fetch(Ey, r1); fetch(Ez, r2); fetch(Bx, r3); fetch(By, r4); fetch(Bz, r5);
        put(0, r0); put(1, r1); put(2, r2); put(3, r3); put(4, r4); put(5, r5);
    }
    __syncthreads();
    GPROF_CLOCK(prof_t1);
    GPROF_ADD(0, prof_t1 - prof_t0);
    GPROF_ADD(2, 1);

    int claimed = 0;
    // next trip: static shares, or the chunk claimed a trip ago (lane 0 is in the loop whenever any lane of the wave is:
    // it holds the chunk's first particle)
    auto advance = [&]() {
        ip = cb_next + lane;
        cb_next = start + 64 * __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
    };
    for (; ip < end; advance()) {
        GPROF_CLOCK(prof_a);
        double xp = nxt[0], yp = nxt[1], zp = nxt[2], ux0, uy0, uz0;
        asm volatile("" ::: "memory");
        ux0 = p.ux[ip]; uy0 = p.uy[ip]; uz0 = p.uz[ip];   // requested now, used after the gather
        asm volatile("" ::: "memory");
        if (lane == 0) claimed = atomicAdd(&next_chunk, 1);
        if (cb_next + lane < end) load_particle(cb_next + lane);
#ifdef WXA_GATHER_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xp), "+v"(yp), "+v"(zp));
#endif
        GPROF_CLOCK(prof_b);
        GatherShapes<O, G> s;
        gather_shapes<O, G>(xp, yp, zp, g, s);
        // staged range check on the extreme points of the node / cell stencils
        constexpr int NN = O + 1, NC = O + 1 - G;
        const int lo_i = min(s.jn, s.jc) - o0, hi_i = max(s.jn + NN, s.jc + NC) - 1 - o0;
        const int lo_j = min(s.kn, s.kc) - o1, hi_j = max(s.kn + NN, s.kc + NC) - 1 - o1;
        const int lo_k = min(s.ln, s.lc) - o2, hi_k = max(s.ln + NN, s.lc + NC) - 1 - o2;
        bool staged = lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < N && hi_j < N && hi_k < N;
        if constexpr (SPLIT) {   // ... and the nodal stencils inside the eleven points -1 .. 9 (o = the tile's first cell - 2)
            staged = staged && s.jn - o0 >= 1 && s.kn - o1 >= 1 && s.ln - o2 >= 1;
        }
        if (!staged) {
            push_straggler(ip);   // stencil leaves the staged tile: handled by gather_push_stragglers_kernel
            continue;
        }
        const int jn = s.jn - o0, jc = s.jc - o0, kn = s.kn - o1, kc = s.kc - o1, ln = s.ln - o2, lc = s.lc - o2;
        double Exp, Eyp, Ezp, Bxp, Byp, Bzp;
        if constexpr (SPLIT) {
            // component c's array: n(c, d) points from lo(c, d) along d; jn .. lc above count from -2
            auto rows = [&](auto cc, auto nx, auto ny, auto nz, const int i, const int j, const int k, const double* sx_,
                            const double* sy_, const double* sz_) {
                constexpr int c = decltype(cc)::value, n0 = SplitTile::n(c, 0), n1 = SplitTile::n(c, 1);
                const double* b_ = F + SplitTile::off(c) + (i - (SplitTile::lo(c, 0) + 2)) +
                                   n0 * ((j - (SplitTile::lo(c, 1) + 2)) + n1 * (k - (SplitTile::lo(c, 2) + 2)));
                return gather_rows_lds<decltype(nx)::value, decltype(ny)::value, decltype(nz)::value, n0, n0 * n1, 2>(b_, sx_, sy_, sz_);
            };
            using std::integral_constant;
            constexpr integral_constant<int, NN> nn{};
            constexpr integral_constant<int, NC> nc{};
            Exp = rows(integral_constant<int, 0>{}, nc, nn, nn, jc, kn, ln, s.sxc, s.syn, s.szn);
            Eyp = rows(integral_constant<int, 1>{}, nn, nc, nn, jn, kc, ln, s.sxn, s.syc, s.szn);
            Ezp = rows(integral_constant<int, 2>{}, nn, nn, nc, jn, kn, lc, s.sxn, s.syn, s.szc);
            Bzp = rows(integral_constant<int, 5>{}, nc, nc, nn, jc, kc, ln, s.sxc, s.syc, s.szn);
            Byp = rows(integral_constant<int, 4>{}, nc, nn, nc, jc, kn, lc, s.sxc, s.syn, s.szc);
            Bxp = rows(integral_constant<int, 3>{}, nn, nc, nc, jn, kc, lc, s.sxn, s.syc, s.szc);
        } else {
#define GROWS(...)                                                                                         \
    [&](const double* b_, const double* sx_, const double* sy_, const double* sz_) {                       \
        /* two rows in flight ahead of the fma chain (order 4: rows of five points -- the compiler's own LDS reads; the */ \
        /* hand-placed waits count rows of <= 4) */                                                                    \
        if constexpr (O <= 3) return gather_rows_lds<__VA_ARGS__, N, N * N, 2>(b_, sx_, sy_, sz_);        \
        else return gather_rows<__VA_ARGS__>(b_, N, N * N, sx_, sy_, sz_);                                 \
    }
            Exp = GROWS(NC, NN, NN)(F + 0 * NPTS + jc + N * (kn + N * ln), s.sxc, s.syn, s.szn);
            Eyp = GROWS(NN, NC, NN)(F + 1 * NPTS + jn + N * (kc + N * ln), s.sxn, s.syc, s.szn);
            Ezp = GROWS(NN, NN, NC)(F + 2 * NPTS + jn + N * (kn + N * lc), s.sxn, s.syn, s.szc);
            Bzp = GROWS(NC, NC, NN)(F + 5 * NPTS + jc + N * (kc + N * ln), s.sxc, s.syc, s.szn);
            Byp = GROWS(NC, NN, NC)(F + 4 * NPTS + jc + N * (kn + N * lc), s.sxc, s.syn, s.szc);
            Bxp = GROWS(NN, NC, NC)(F + 3 * NPTS + jn + N * (kc + N * lc), s.sxn, s.syc, s.szc);
#undef GROWS
        }
#ifdef WXA_GATHER_PROFILE
        double prof_e = Exp + Bxp;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(prof_e));
#endif
        GPROF_CLOCK(prof_c);
        push_and_store<PUSHER, MOVE>(p, ip, xp, yp, zp, ux0, uy0, uz0, Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, m, dt, ext, hook,
                                     count_local ? lhist : nullptr, tile);
#ifdef WXA_GATHER_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GPROF_CLOCK(prof_d);
        GPROF_ADD(3, prof_b - prof_a); GPROF_ADD(4, prof_c - prof_b); GPROF_ADD(5, prof_d - prof_c); GPROF_ADD(6, 1);
        if (prof_e == 1.2345e-300) p.x[ip] = prof_e;
#endif
    }
#ifdef WXA_GATHER_PROFILE
    { GPROF_CLOCK(prof_t2); GPROF_ADD(1, prof_t2 - prof_t1); }
#endif
    {
        __syncthreads();
        const int n = min(sn, SCAP);
        if (tid == 0 && n > 0) sbase = (int)atomicAdd(sq.count, (unsigned)n);
        __syncthreads();
        for (int i = tid; i < n; i += GT_THREADS) sq.idx[sbase + i] = slist[i];
    }
    if constexpr (MOVE) {
        if (count_local) {
            __syncthreads();
            push_sort_tile_finish(hook, lhist, tile, tid);
        }
    }
}

template <int O, int G, int PUSHER, bool MOVE>
__global__ void __launch_bounds__(256)
gather_push_stragglers_kernel(PV p, const int* __restrict__ idx, const unsigned* __restrict__ count, DevF Ex,
                              DevF Ey, DevF Ez, DevF Bx, DevF By, DevF Bz, Geom g, double q, double m, double dt,
                              ExtEB ext, PushSort hook) {
    const unsigned n = *count;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int ip = idx[t];
        const double xp = p.x[ip], yp = p.y[ip], zp = p.z[ip];
        GatherShapes<O, G> s;
        gather_shapes<O, G>(xp, yp, zp, g, s);
        double Exp, Eyp, Ezp, Bxp, Byp, Bzp;
        gather_global<O, G>(s, Ex, Ey, Ez, Bx, By, Bz, Exp, Eyp, Ezp, Bxp, Byp, Bzp);
        push_and_store<PUSHER, MOVE>(p, ip, xp, yp, zp, p.ux[ip], p.uy[ip], p.uz[ip], Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, m, dt,
                                     ext, hook);
    }
}

bool gather_tile_available(const wxa_workspace* ws, const wxa_particle_view* p) {
    return ws && ws->sorted_valid && ws->sorted_x == p->x && ws->sorted_np <= p->np;
}

template <int PUSHER, bool MOVE, int PART = 0>
static wxa_status launch(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                         const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                         wxa_workspace* ws, hipStream_t st) {
    GTileGeom tg;
    for (int d = 0; d < 3; ++d) {
        tg.nt[d] = (ws->sort_nc[d] + GT_TS - 1) / GT_TS;
        tg.cell_lo[d] = ws->sort_cell_lo[d];
    }
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const PV pv = make_pv(*p);
    const Geom g = make_geom(*geom);
    const int* offsets = (const int*)ws->offsets.p;
    const DevF ex = make_devf(E[0]), ey = make_devf(E[1]), ez = make_devf(E[2]);
    const DevF bx = make_devf(B[0]), by = make_devf(B[1]), bz = make_devf(B[2]);
    wxa_status rc;
    if ((rc = ws->stragglers.reserve(sizeof(int) * (size_t)p->np + 64)) != WXA_OK) return rc;
    GatherStragglers sq{(int*)ws->stragglers.p, nullptr, nullptr};
    unsigned *cnt_now = nullptr, *cnt_next = nullptr;
    if ((rc = flip_counter(ws, 16, ws->gather_flips, st, cnt_now, cnt_next)) != WXA_OK) return rc;
    sq.count = cnt_now; sq.next = cnt_next;   // words 16, 17 of ws->counters
    const ExtEB ext = ext_of(ws);
    const PushSort hook = make_push_sort(ws, 0, MOVE);
    HeavyUnits hu;
    long extra_groups = 0;
    if ((rc = plan_heavy_tiles(ws, offsets, ntiles, (long)p->np, hu, extra_groups, st)) != WXA_OK) return rc;
    const dim3 grid((unsigned)(xcd_grid_size(ntiles) + extra_groups)), block(GT_THREADS);
#define WXA_GT(O, G)                                                                                        \
    do {                                                                                                    \
        hipLaunchKernelGGL((gather_push_tile_kernel<O, G, PUSHER, MOVE, PART>), grid, block, 0, st, pv, offsets, ex, \
                           ey, ez, bx, by, bz, g, tg, q, m, dt, sq, ext, hook, hu);                                   \
        hipLaunchKernelGGL((gather_push_stragglers_kernel<O, G, PUSHER, MOVE>), dim3(WXA_STRAGGLER_BLOCKS), dim3(256), 0, st, pv, \
                           sq.idx, sq.count, ex, ey, ez, bx, by, bz, g, q, m, dt, ext, hook);                     \
    } while (0)
    if (galerkin) {
        if (order == 1) WXA_GT(1, 1); else if (order == 2) WXA_GT(2, 1); else if (order == 3) WXA_GT(3, 1); else WXA_GT(4, 1);
    } else {
        if (order == 1) WXA_GT(1, 0); else if (order == 2) WXA_GT(2, 0); else if (order == 3) WXA_GT(3, 0); else WXA_GT(4, 0);
    }
#undef WXA_GT
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status gather_push_tiled(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                             const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                             int pusher, bool move, wxa_workspace* ws, hipStream_t st) {
    if (pusher == WXA_PUSHER_BORIS) {
        if (move) return launch<WXA_PUSHER_BORIS, true>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_BORIS, false>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (pusher == WXA_PUSHER_VAY) {
        if (move) return launch<WXA_PUSHER_VAY, true>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_VAY, false>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (pusher == WXA_PUSHER_HC) {
        if (move) return launch<WXA_PUSHER_HC, true>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_HC, false>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (move) return launch<WXA_PUSHER_BORIS_RR, true>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    return launch<WXA_PUSHER_BORIS_RR, false>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
}

// PushPX on one part of the tiles (part = 1 interior, 2 faces), see the kernel
wxa_status gather_push_tiled_part(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                                  const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                                  int pusher, int part, wxa_workspace* ws, hipStream_t st) {
    if (pusher == WXA_PUSHER_BORIS) {
        if (part == 1) return launch<WXA_PUSHER_BORIS, true, 1>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_BORIS, true, 2>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (pusher == WXA_PUSHER_VAY) {
        if (part == 1) return launch<WXA_PUSHER_VAY, true, 1>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_VAY, true, 2>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (pusher == WXA_PUSHER_HC) {
        if (part == 1) return launch<WXA_PUSHER_HC, true, 1>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
        return launch<WXA_PUSHER_HC, true, 2>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    }
    if (part == 1) return launch<WXA_PUSHER_BORIS_RR, true, 1>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
    return launch<WXA_PUSHER_BORIS_RR, true, 2>(p, E, B, geom, q, m, dt, order, galerkin, ws, st);
}

}  // namespace wxa

#ifdef WXA_GATHER_PROFILE
extern "C" int wxa_debug_gather_profile(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wxa::wxa_gather_prof), sizeof(wxa::wxa_gather_prof)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wxa::wxa_gather_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
