// LDS-tile current deposition for gfx950.
//
// The reference's own shared-memory deposition (doDepositionSharedShapeN,
// Source/Particles/Deposition/CurrentDeposition.H:453-616) handles only the direct scheme,
// one component per pass.  Here one workgroup owns one tile of 8x8x8 cells of the
// tile-major cell sort (wxa_sort_particles_by_cell), keeps all three J components of the
// tile plus its stencil halo (and one point of drift margin) in LDS as fp64
// (3 x 15 x 232 x 8 B = 83.5 KB), accumulates with ds_add_f64, and writes each non-zero LDS
// point back to HBM with one global fp64 atomic.
//
// Per trip the workgroup stages up to 1024 particles through LDS (coalesced, 56 KB), keys them by
// stencil frame, and builds work items: two neighbours of one cell that do not cross a cell form
// a PAIR (merged in registers before the atomics), a particle that crosses a cell is a slow
// single.  At most 512 fast items are taken per trip -- one per lane, so the pass over them is a
// single pass for every wave -- and each goes to the lane (quarter-wave, LDS bank of its frame
// base): the 16 lanes that a ds_add_f64 serves together then always hit 16 different banks.
// Slow singles are deferred per tile and run once through the general Esirkepov body, 64 to a
// wave, the three components on different waves.  Particles whose stencil leaves the LDS tile
// (drift of more than a cell since the last sort, particles outside the domain before the
// periodic wrap) are queued and deposited with global atomics by a second kernel, so correctness
// never depends on the sort being fresh.  DESIGN.md section 3 has the measurements behind each step.
#include "deposit_body.hpp"
#include "workspace.hpp"

namespace wxa {

constexpr int TS = WXA_TILE;          // tile edge in cells (workspace.hpp)
constexpr int TILE_CELLS = TS * TS * TS;

template <int M>
struct TileDims {
    static constexpr int LO = -2 - M;            // first LDS point relative to the tile's first cell
    static constexpr int N = TS + 5 + 2 * M;     // points per direction
    // Plane stride padded to 8 (mod 16): with the row stride N = 15 = -1 (mod 16) the LDS bank of
    // point (i,j,k) is (i - j + 8k) mod 16, so the 16 cells {8 i} x {k, k+1} of one row j sit on
    // 16 different banks (see the lane assignment in the kernel and cell_of in particles.hip).
    static constexpr int PS = N * N + ((8 - (N * N) % 16) + 16) % 16;
    static constexpr int NPTS = N * PS;          // doubles per component (incl. padding)
};

template <int M>
struct LdsSink {
    double* base;   // LDS address of slot 0 of component 0: every deposit is base + a compile-time offset
    __device__ __forceinline__ LdsSink(double* lds, int oi, int oj, int ok)
        : base(lds + oi + TileDims<M>::N * oj + TileDims<M>::PS * ok) {}
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        constexpr int N = TileDims<M>::N;
        atomic_add_f64(base + (c * TileDims<M>::NPTS + i + N * j + TileDims<M>::PS * k), v);
    }
    __device__ __forceinline__ void add_abs(int c, int gi, int gj, int gk, double v) { add(c, gi, gj, gk, v); }
};

// Phase clocks of the tile kernel (opt-in: WXA_DEPOSIT_PROFILE=1 python -m warpx_amd.build --force;
// read with scripts/deposit_profile.py).  Thread 0 of every workgroup accumulates the cycles
// between marks; a barrier before the mark makes the interval the workgroup's, not wave 0's.
#ifdef WXA_DEPOSIT_PROFILE
__device__ unsigned long long wxa_dep_prof[16];
// accumulated in registers of thread 0, added to the global counters once per workgroup (a global
// atomic per mark would serialise on the 16 addresses and distort the very thing measured)
#define DPROF_INIT                                  \
    long long prof_t = clock64();                   \
    unsigned long long prof_acc[16];                \
    _Pragma("unroll") for (int prof_i = 0; prof_i < 16; ++prof_i) prof_acc[prof_i] = 0;
#define DPROF(n)                                                  \
    do {                                                          \
        __syncthreads();                                          \
        const long long prof_n = clock64();                       \
        prof_acc[n] += (unsigned long long)(prof_n - prof_t);     \
        prof_t = prof_n;                                          \
    } while (0)
#define DCOUNT(n, v) prof_acc[n] += (unsigned long long)(v)
#define DPROF_FINISH                                                                        \
    if (threadIdx.x == 0) {                                                                 \
        _Pragma("unroll") for (int prof_i = 0; prof_i < 16; ++prof_i)                       \
            if (prof_acc[prof_i]) atomicAdd(&wxa_dep_prof[prof_i], prof_acc[prof_i]);       \
    }
#else
#define DPROF_INIT
#define DPROF(n)
#define DCOUNT(n, v)
#define DPROF_FINISH
#endif

struct TileGeom {
    int nt[3];        // tiles per direction
    int cell_lo[3];   // global index of the brick's first cell
};

constexpr int DT_THREADS = 512;   // 8 waves: one workgroup per CU (LDS-limited), 2 waves per SIMD
constexpr int DT_BATCH = 1024;    // particles staged in LDS per round (7 x 1024 x 8 B = 56 KB)
constexpr int NBANK = 16;         // LDS banks (8-byte wide) seen by one step of a ds_add_f64
constexpr int ROWS = 512 / NBANK; // quarter-waves of a workgroup = rows of the lane-assignment table
constexpr int DT_DEFER = 1024;    // capacity of the per-tile list of deferred (cell-crossing) particles

// Particles whose stencil leaves the LDS tile (stale sort, particles outside the domain before
// the periodic wrap) are queued and deposited by deposit_stragglers_kernel with global atomics:
// keeping that path out of the tile kernel saves registers and instruction cache.
struct StragglerQueue {
    int* __restrict__ idx;
    unsigned* __restrict__ count;
    __device__ __forceinline__ void push(int ip) const { idx[atomicAdd(count, 1u)] = ip; }
};

template <int O, int ALGO, int M>
__global__ void __launch_bounds__(DT_THREADS)
deposit_tile_kernel(const double* __restrict__ px, const double* __restrict__ py,
                    const double* __restrict__ pz, const double* __restrict__ pw,
                    const double* __restrict__ pux, const double* __restrict__ puy,
                    const double* __restrict__ puz, const int* __restrict__ offsets, DevF Jx, DevF Jy,
                    DevF Jz, Geom g, TileGeom tg, double q, double dt, double relative_time,
                    StragglerQueue sq) {
    constexpr int N = TileDims<M>::N;
    constexpr int NPTS = TileDims<M>::NPTS;
    constexpr int PS = TileDims<M>::PS;
    constexpr int PAIRED = 1 << 30;
    __shared__ double lds[3 * NPTS];
    __shared__ double stage[7][DT_BATCH];
    __shared__ int keys[DT_BATCH + 1];
    __shared__ int items[DT_BATCH];
    __shared__ int nitems;
    __shared__ int segcnt[2][DT_BATCH / 64];   // fast / slow items per 64-particle segment
    __shared__ int cut_a, cut_nslow;           // set by the thread holding the first fast item beyond the cap
    __shared__ int pf_scratch[64];             // landing zone of the L2 prefetch loads
    __shared__ int slots[DT_THREADS];          // fast item of every lane (row = quarter-wave, column = LDS bank)
    __shared__ int bcnt[NBANK], novf;
    __shared__ unsigned deferred[DT_DEFER];   // particles with a cell crossing (global indices), kept for one dense pass
    __shared__ int ndeferred;
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const long tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int start = offsets[tile * TILE_CELLS];
    const int end = offsets[(tile + 1) * TILE_CELLS];
    if (end <= start) return;
    const int tid = threadIdx.x;
    DPROF_INIT
    for (int a = tid; a < 3 * NPTS; a += DT_THREADS) lds[a] = 0.0;
    const int ti = (int)(tile % tg.nt[0]);
    const int tj = (int)((tile / tg.nt[0]) % tg.nt[1]);
    const int tk = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
    // global grid index of LDS point 0
    const int o0 = tg.cell_lo[0] + ti * TS + TileDims<M>::LO;
    const int o1 = tg.cell_lo[1] + tj * TS + TileDims<M>::LO;
    const int o2 = tg.cell_lo[2] + tk * TS + TileDims<M>::LO;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) ndeferred = 0;
    constexpr bool ESIRKEPOV = ALGO == WXA_DEPOSIT_ESIRKEPOV;
    constexpr int ROUNDS = DT_BATCH / DT_THREADS;   // particles per thread and batch
    constexpr int WAVES = DT_THREADS / 64;
    constexpr int NSEG = DT_BATCH / 64;             // 64-particle segments of a batch (one ballot each)
    constexpr int FAST_CAP = DT_THREADS;            // one fast item per lane: the fast pass is ONE pass
    const double* const parr[7] = {px, py, pz, pw, pux, puy, puz};
    int nb_cap = DT_BATCH;   // particles staged per batch; shrinks when the item cap cuts batches short
    int b0 = start;
    // One extra trip after the last batch only flushes the deferred list, so that the (large)
    // general-path code exists once in the kernel.
    for (;;) {
        const bool last = b0 >= end;
        if constexpr (!ESIRKEPOV) {
            if (last) break;
        }
        const int nb = last ? 0 : min(nb_cap, end - b0);
        __syncthreads();   // previous round's readers are done (and the zero fill on round 0)
        // ---- stage the batch (coalesced, all loads in flight together) and key every particle
        //      by its stencil frame ----
        int key_r[ROUNDS];     // stencil frame of this thread's particles (< 0: none / straggler)
        bool cross_r[ROUNDS];  // the particle crosses a cell during the step
        {
            double r[ROUNDS][7];
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) {
                const int a = tid + rr * DT_THREADS;
                key_r[rr] = -1; cross_r[rr] = false;
                if (a < nb) {
#pragma unroll
                    for (int c = 0; c < 7; ++c) r[rr][c] = parr[c][b0 + a];
                }
            }
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) {
                const int a = tid + rr * DT_THREADS;
                if (a < nb) {
                    const ParticleState p{r[rr][0], r[rr][1], r[rr][2], r[rr][3], r[rr][4], r[rr][5], r[rr][6]};
#pragma unroll
                    for (int c = 0; c < 7; ++c) stage[c][a] = r[rr][c];
                    int key;
                    if constexpr (ESIRKEPOV) {
                        int bi, bj, bk;
                        const bool cross = esirkepov_frame_cross<O>(p, g, dt, relative_time, bi, bj, bk);
                        const int li = bi - o0, lj = bj - o1, lk = bk - o2;
                        const bool in = li >= 0 && lj >= 0 && lk >= 0 && li + O + 3 <= N && lj + O + 3 <= N &&
                                        lk + O + 3 <= N;
                        // outside the LDS tile: straggler (a key no neighbour shares); queued below,
                        // once it is known that this batch consumes the particle
                        key = in ? (li | (lj << 8) | (lk << 16)) : -2 - a;
                        cross_r[rr] = cross;
                    } else {
                        DirectShapes<O> sh;
                        direct_shapes<O>(p, g, q, relative_time, sh);
                        const int lo_i = min(sh.jn, sh.jc) - o0, lo_j = min(sh.kn, sh.kc) - o1,
                                  lo_k = min(sh.ln, sh.lc) - o2;
                        const int hi_i = max(sh.jn, sh.jc) - o0 + O, hi_j = max(sh.kn, sh.kc) - o1 + O,
                                  hi_k = max(sh.ln, sh.lc) - o2 + O;
                        key = (lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < N && hi_j < N && hi_k < N) ? 0 : -1;
                        if (key < 0) sq.push(b0 + a);
                    }
                    keys[a] = key;
                    key_r[rr] = key;
                }
            }
        }
        if (tid == 0) { nitems = 0; cut_a = nb; cut_nslow = -1; novf = 0; }
        if (tid < NBANK) bcnt[tid] = 0;
        if constexpr (!ESIRKEPOV) {
            __syncthreads();
            DPROF(0);
        }
        if constexpr (ESIRKEPOV) {
            // Nobody appends to the deferred list before the flush decision below.
            const int nd0 = ndeferred;
            // ---- work items.  A run of equal frames (the particles of one cell, cut at the
            // 64-particle segments) is split into pairs (+ one single if odd); an item is "slow" if
            // one of its particles crosses a cell.  Ranks come from ballots and a 16-entry prefix,
            // so both lists keep the cell order of the sort (neighbouring lanes -> neighbouring
            // LDS addresses).  At most FAST_CAP fast items are taken: the batch ends where the
            // next one would start (a_cut) and the rest is staged again by the next trip.
            bool is_item[ROUNDS], is_slow[ROUNDS], is_pair[ROUNDS];
            int rank_f[ROUNDS], rank_s[ROUNDS];
            const unsigned long long le = ~0ull >> (63 - lane), lt = le >> 1;
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) {
                // the neighbours' keys come from the neighbouring lanes (runs are cut at the wave's
                // 64 particles anyway), so this needs no barrier after the staging
                const int key = key_r[rr];
                const bool cr = cross_r[rr];
                const int kprev = __shfl_up(key, 1), knext = __shfl_down(key, 1);
                const bool cprev = __shfl_up((int)cr, 1) != 0, cnext = __shfl_down((int)cr, 1) != 0;
                // a crossing particle is a run of its own (slow single); pairs form among the others
                const bool head = lane == 0 || kprev != key || cr || cprev;
                const unsigned long long H = __ballot(head);
                const int run_start = 63 - __clzll((long long)(H & le));   // lane 0 is always a head
                const bool it = key >= 0 && ((lane - run_start) & 1) == 0;
                const bool pr = it && !cr && lane < 63 && knext == key && !cnext;
                const bool sl = it && cr;
                const unsigned long long F = __ballot(it && !sl), S = __ballot(sl);
                rank_f[rr] = __popcll(F & lt);
                rank_s[rr] = __popcll(S & lt);
                if (lane == 0) {
                    segcnt[0][wave + rr * WAVES] = __popcll(F);
                    segcnt[1][wave + rr * WAVES] = __popcll(S);
                }
                is_item[rr] = it; is_slow[rr] = sl; is_pair[rr] = pr;
            }
            __syncthreads();
            DPROF(0);   // zero fill (first trip) + stage + key + item flags
            int tot_f = 0, tot_s = 0;
            {
                int pre_f[ROUNDS], pre_s[ROUNDS];
#pragma unroll
                for (int rr = 0; rr < ROUNDS; ++rr) pre_f[rr] = pre_s[rr] = 0;
#pragma unroll
                for (int sg = 0; sg < NSEG; ++sg) {
                    const int cf = segcnt[0][sg], cs = segcnt[1][sg];
#pragma unroll
                    for (int rr = 0; rr < ROUNDS; ++rr)
                        if (sg < wave + rr * WAVES) { pre_f[rr] += cf; pre_s[rr] += cs; }
                    tot_f += cf; tot_s += cs;
                }
                // Lane assignment of the fast items.  ds_add_f64 serves 16 lanes (a quarter-wave) per
                // step from 16 banks of 8 bytes (scripts/microbench/lds_*_bench.hip), and ANY two
                // lanes on one bank double the cost of the step; every deposit of a lane is (frame
                // base + compile-time offset), so lanes conflict exactly when their frame bases share
                // a bank.  Items are therefore bucketed by the bank of their frame base: column =
                // bank, row (= quarter-wave) = rank in the bucket.  All items of a cell land in one
                // column, so two lanes of a quarter-wave never hit the same address either.
#pragma unroll
                for (int rr = 0; rr < ROUNDS; ++rr) {
                    const int a = tid + rr * DT_THREADS;
                    const int e = a | (is_pair[rr] ? PAIRED : 0);
                    const int rank = pre_f[rr] + rank_f[rr];
                    const bool fast = is_item[rr] && !is_slow[rr] && rank < FAST_CAP;
                    const int kk = key_r[rr];
                    const int bank = fast ? ((kk & 255) + N * ((kk >> 8) & 255) + PS * (kk >> 16)) & (NBANK - 1) : -1;
                    // (a wave-aggregated rank -- 16 ballots, one atomic per wave and bank -- was slower)
                    const int row = fast ? atomicAdd(&bcnt[bank], 1) : 0;
                    if (fast) {
                        if (row < ROWS) slots[row * NBANK + bank] = e;
                        else items[atomicAdd(&novf, 1)] = e;   // bucket full: takes a free slot below
                    }
                    if (is_item[rr]) {
                        if (is_slow[rr]) items[DT_BATCH - 1 - (pre_s[rr] + rank_s[rr])] = e;   // from the back
                        else if (rank == FAST_CAP) { cut_a = a; cut_nslow = pre_s[rr] + rank_s[rr]; }
                    }
                }
            }
            __syncthreads();
            DPROF(1);   // work-item lists
            const int a_cut = cut_a;                              // particles consumed by this trip
            const int nfast = min(tot_f, FAST_CAP);
            const int nsl = cut_nslow >= 0 ? cut_nslow : tot_s;   // slow items before a_cut
            if (a_cut < nb) nb_cap = min(DT_BATCH, ((a_cut + 127) >> 6) << 6);
            else if (nb == nb_cap && nb_cap < DT_BATCH && tot_f < FAST_CAP - 32) nb_cap += 64;
#pragma unroll
            for (int rr = 0; rr < ROUNDS; ++rr) {
                const int a = tid + rr * DT_THREADS;
                if (a < a_cut && key_r[rr] <= -2) sq.push(b0 + a);
            }
            // Pull the next batch towards the L2 while this one is deposited: one 4-byte load per
            // 128-byte line, straight into an LDS scratch word (no register, no wait until the
            // next barrier).  Wave w touches array w.
            if (!last && wave < 7) {
                const int nx0 = b0 + a_cut;
                const int nxn = min(nb_cap, end - nx0);
                if (lane * 16 < nxn)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(parr[wave] + nx0 + lane * 16),
                        (__attribute__((address_space(3))) void*)pf_scratch, 4, 0, 0);
            }
            DCOUNT(15, novf);
            // This lane's fast item: slot (row, column) = (quarter-wave, bank) of the table if its
            // bucket reaches this row, else one of the items of overfull buckets.  Free slots are
            // numbered column by column, so that consecutive overflow items (same bucket, often the
            // same cell) land in different quarter-waves: each one then conflicts with a single lane.
            int e;
            {
                const int col = tid & (NBANK - 1), row = tid / NBANK;
                if (row < bcnt[col]) {
                    e = slots[tid];
                } else {
                    int m = row - bcnt[col];
#pragma unroll
                    for (int c2 = 0; c2 < NBANK; ++c2)
                        if (c2 < col) m += ROWS - min(bcnt[c2], ROWS);
                    e = m < novf ? items[m] : -1;
                }
            }
            if (e >= 0) {   // pairs (and singles) that stay in their cell
                const int a = e & (PAIRED - 1);
                const bool paired = (e & PAIRED) != 0;
                const int a2 = paired ? a + 1 : a;
                const ParticleState p1{stage[0][a], stage[1][a], stage[2][a], stage[3][a],
                                       stage[4][a], stage[5][a], stage[6][a]};
                const ParticleState p2{stage[0][a2], stage[1][a2], stage[2][a2], stage[3][a2],
                                       stage[4][a2], stage[5][a2], stage[6][a2]};
                EsirkepovNC<O> s1, s2;
                esirkepov_nc_shapes<O>(p1, g, q, dt, relative_time, s1);
                esirkepov_nc_shapes<O>(p2, g, q, dt, relative_time, s2);
                const int key = keys[a];
                LdsSink<M> sink(lds, key & 255, (key >> 8) & 255, (key >> 16) & 255);
                esirkepov_accumulate_pair_nc<O>(s1, s2, /*null2=*/!paired, g, dt, sink);
            }
            DPROF(2);   // fast pass
            // Particles with a cell crossing take the general path (alone: merging two of them would
            // double its already long instruction stream), whose cost per wave does not depend on how
            // many lanes are active.  A handful per batch (thermal plasma: ~1.5 %) would cost every
            // batch a full pass, so they are deferred (as global particle indices) and run packed
            // 64 to a wave, the three J components on different waves: when the list would overflow
            // (hot / relativistic plasma: every batch) and once at the end of the tile.
            int nd_base = nd0;
            DCOUNT(8, nfast); DCOUNT(9, nsl); DCOUNT(10, 1); DCOUNT(11, a_cut); DCOUNT(12, nb);
            if (last || nd0 + nsl > DT_DEFER) {   // block-uniform
                DCOUNT(13, 1); DCOUNT(14, nd0);
                // work unit = (64 deferred particles, one J component); units go round the waves
                const int nunits = 3 * ((nd0 + 63) >> 6);
                for (int u = wave; u < nunits; u += WAVES) {
                    const int it = (u / 3) * 64 + lane;
                    if (it >= nd0) continue;
                    const int ip = (int)deferred[it];
                    const ParticleState p1{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
                    EsirkepovShapes<O> s1;
                    esirkepov_shapes<O>(p1, g, q, dt, relative_time, s1);
                    LdsSink<M> sink(lds, s1.bi - o0, s1.bj - o1, s1.bk - o2);
                    switch (u % 3) {
                        case 0: esirkepov_accumulate_comp<O, 0>(s1, g, dt, sink); break;
                        case 1: esirkepov_accumulate_comp<O, 1>(s1, g, dt, sink); break;
                        default: esirkepov_accumulate_comp<O, 2>(s1, g, dt, sink); break;
                    }
                }
                nd_base = 0;
                __syncthreads();   // the list is free again
            }
            DPROF(3);   // deferred general-path flush
            for (int rk = tid; rk < nsl; rk += DT_THREADS) {
                const int e = items[DT_BATCH - 1 - rk];
                deferred[nd_base + rk] = (unsigned)(b0 + (e & (PAIRED - 1)));   // slow items are singles
            }
            if (tid == 0) ndeferred = nd_base + nsl;
            if (last) break;
            b0 += a_cut;
        } else {
            for (int a = tid; a < nb; a += DT_THREADS)
                if (keys[a] >= 0) items[atomicAdd(&nitems, 1)] = a;
            __syncthreads();
            DPROF(1);
            constexpr int IPC = 4;
            const int total = nitems;
            const int nchunks = ((total + 64 * IPC - 1) / (64 * IPC)) * IPC;
            for (int c = wave; c < nchunks; c += WAVES) {
                const int it = (c / IPC) * (64 * IPC) + lane * IPC + (c % IPC);
                if (it >= total) continue;
                const int a = items[it];
                const ParticleState p1{stage[0][a], stage[1][a], stage[2][a], stage[3][a],
                                       stage[4][a], stage[5][a], stage[6][a]};
                DirectShapes<O> sh;
                direct_shapes<O>(p1, g, q, relative_time, sh);
                LdsSink<M> sink(lds, -o0, -o1, -o2);
                direct_accumulate<O>(sh, sink);
            }
            b0 += nb;
        }
    }
    __syncthreads();
    DPROF(4);   // append to the deferred list / direct pass
    // write-back: one global atomic per non-zero LDS point (tiles overlap on their halos)
    const DevF* Jc[3] = {&Jx, &Jy, &Jz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const DevF& J = *Jc[c];
        for (int a = tid; a < NPTS; a += DT_THREADS) {
            const double v = lds[c * NPTS + a];
            if (v != 0.0) {
                // a = i + N j + PS k; the padding words of a plane (a % PS >= N N) stay zero
                const int i = o0 + (a % PS) % N, j = o1 + (a % PS) / N, k = o2 + a / PS;
                if (i >= J.lo0 && i < J.lo0 + J.n0 && j >= J.lo1 && j < J.lo1 + J.n1 && k >= J.lo2 &&
                    k < J.lo2 + J.n2)
                    atomic_add_f64(J.p + J.off(i, j, k), v);
            }
        }
    }
    DPROF(5);   // write-back
    DPROF_FINISH
}

template <int O, int ALGO>
__global__ void __launch_bounds__(256)
deposit_stragglers_kernel(const double* __restrict__ px, const double* __restrict__ py,
                          const double* __restrict__ pz, const double* __restrict__ pw,
                          const double* __restrict__ pux, const double* __restrict__ puy,
                          const double* __restrict__ puz, const int* __restrict__ idx,
                          const unsigned* __restrict__ count, DevF Jx, DevF Jy, DevF Jz, Geom g, double q,
                          double dt, double relative_time) {
    const unsigned n = *count;
    GlobalSink gs = make_global_sink(Jx, Jy, Jz);
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int ip = idx[t];
        const ParticleState p{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
        if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
            EsirkepovShapes<O> s;
            esirkepov_shapes<O>(p, g, q, dt, relative_time, s);
            gs.bi = s.bi; gs.bj = s.bj; gs.bk = s.bk;
            esirkepov_accumulate<O>(s, g, dt, gs);
        } else {
            DirectShapes<O> s;
            direct_shapes<O>(p, g, q, relative_time, s);
            direct_accumulate<O>(s, gs);
        }
    }
}

bool deposit_tile_available(const wxa_workspace* ws, const wxa_particle_view* p) {
    return ws && ws->sorted_valid && ws->sorted_x == p->x && ws->sorted_np <= p->np;
}

template <int O, int ALGO>
static wxa_status launch_tile(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* geom,
                              double q, double dt, double relative_time, wxa_workspace* ws, hipStream_t st) {
    TileGeom tg;
    for (int d = 0; d < 3; ++d) {
        tg.nt[d] = (ws->sort_nc[d] + TS - 1) / TS;
        tg.cell_lo[d] = ws->sort_cell_lo[d];
    }
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const Geom g = make_geom(*geom);
    const int* offsets = (const int*)ws->offsets.p;
    const dim3 grid((unsigned)xcd_grid_size(ntiles)), block(DT_THREADS);
    wxa_status rc;
    if ((rc = ws->stragglers.reserve(sizeof(int) * (size_t)p->np + 64)) != WXA_OK) return rc;
    if ((rc = ws->counters.reserve(256)) != WXA_OK) return rc;
    StragglerQueue sq{(int*)ws->stragglers.p, (unsigned*)ws->counters.p};
    WXA_HIP_CHECK(hipMemsetAsync(sq.count, 0, sizeof(unsigned), st));
    const DevF jx = make_devf(J[0]), jy = make_devf(J[1]), jz = make_devf(J[2]);
    // One extra LDS point per side beyond the exact stencil reach: particles may have drifted up
    // to one cell since the last sort (sort_intervals > 1) before they have to take the
    // straggler path (global atomics, ~7 ns per particle).  3 x 15^3 x 8 B = 81 KB + 56 KB stage.
    constexpr int M = 1;
    hipLaunchKernelGGL((deposit_tile_kernel<O, ALGO, M>), grid, block, 0, st, p->x, p->y, p->z, p->w, p->ux,
                       p->uy, p->uz, offsets, jx, jy, jz, g, tg, q, dt, relative_time, sq);
    hipLaunchKernelGGL((deposit_stragglers_kernel<O, ALGO>), dim3(512), dim3(256), 0, st, p->x, p->y, p->z, p->w,
                       p->ux, p->uy, p->uz, sq.idx, sq.count, jx, jy, jz, g, q, dt, relative_time);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status deposit_current_tiled(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* geom,
                                 double q, double dt, double relative_time, int order, int algo,
                                 wxa_workspace* ws, hipStream_t st) {
    if (algo == WXA_DEPOSIT_ESIRKEPOV) {
        if (order == 1) return launch_tile<1, WXA_DEPOSIT_ESIRKEPOV>(p, J, geom, q, dt, relative_time, ws, st);
        if (order == 2) return launch_tile<2, WXA_DEPOSIT_ESIRKEPOV>(p, J, geom, q, dt, relative_time, ws, st);
        return launch_tile<3, WXA_DEPOSIT_ESIRKEPOV>(p, J, geom, q, dt, relative_time, ws, st);
    }
    if (order == 1) return launch_tile<1, WXA_DEPOSIT_DIRECT>(p, J, geom, q, dt, relative_time, ws, st);
    if (order == 2) return launch_tile<2, WXA_DEPOSIT_DIRECT>(p, J, geom, q, dt, relative_time, ws, st);
    return launch_tile<3, WXA_DEPOSIT_DIRECT>(p, J, geom, q, dt, relative_time, ws, st);
}

}  // namespace wxa

#ifdef WXA_DEPOSIT_PROFILE
extern "C" int wxa_debug_deposit_profile(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wxa::wxa_dep_prof), sizeof(wxa::wxa_dep_prof)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wxa::wxa_dep_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
