// LDS-tile current deposition for gfx950.
//
// The reference's own shared-memory deposition (doDepositionSharedShapeN,
// Source/Particles/Deposition/CurrentDeposition.H:453-616) handles only the direct scheme,
// one component per pass.  Here one workgroup owns one tile of 8x8x8 cells of the
// tile-major cell sort (wxa_sort_particles_by_cell), keeps all three J components of the
// tile plus its stencil halo (and one point of drift margin) in LDS as fp64
// (3 x 15^3 x 8 B = 81 KB), accumulates with ds_add_f64, and writes each non-zero LDS point
// back to HBM with one global fp64 atomic.  Particles are staged through LDS in coalesced
// batches of 1024 (56 KB) and handed to the lanes as neighbouring PAIRS with a strided walk:
// the two particles of a pair usually share the cell (merged before the atomics, halving the
// load on the binding LDS-atomic pipe), while the 64 lanes of a wave work on different cells
// (no same-address serialisation).  Particles whose stencil leaves the LDS tile (drift of more
// than a cell since the last sort, particles outside the domain before the periodic wrap) are
// queued and deposited with global atomics by a second kernel, so correctness never depends
// on the sort being fresh.
#include "deposit_body.hpp"
#include "workspace.hpp"

namespace wxa {

constexpr int TS = WXA_TILE;          // tile edge in cells (workspace.hpp)
constexpr int TILE_CELLS = TS * TS * TS;

template <int M>
struct TileDims {
    static constexpr int LO = -2 - M;            // first LDS point relative to the tile's first cell
    static constexpr int N = TS + 5 + 2 * M;     // points per direction
    static constexpr int NPTS = N * N * N;
};

template <int M>
struct LdsSink {
    double* base;   // LDS address of slot 0 of component 0: every deposit is base + a compile-time offset
    __device__ __forceinline__ LdsSink(double* lds, int oi, int oj, int ok)
        : base(lds + oi + TileDims<M>::N * (oj + TileDims<M>::N * ok)) {}
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        constexpr int N = TileDims<M>::N;
        atomic_add_f64(base + (c * TileDims<M>::NPTS + i + N * (j + N * k)), v);
    }
    __device__ __forceinline__ void add_abs(int c, int gi, int gj, int gk, double v) { add(c, gi, gj, gk, v); }
};

struct TileGeom {
    int nt[3];        // tiles per direction
    int cell_lo[3];   // global index of the brick's first cell
};

constexpr int DT_THREADS = 512;   // 8 waves: one workgroup per CU (LDS-limited), 2 waves per SIMD
constexpr int DT_BATCH = 1024;    // particles staged in LDS per round (7 x 1024 x 8 B = 56 KB)
constexpr int DT_DEFER = 1024;    // capacity of the per-tile list of deferred (cell-crossing) pairs
constexpr int DT_DENSE = 256;     // a batch with this many crossing pairs runs the general path at once

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
    constexpr int PAIRED = 1 << 30;
    __shared__ double lds[3 * NPTS];
    __shared__ double stage[7][DT_BATCH];
    __shared__ int keys[DT_BATCH + 1];
    __shared__ int items[DT_BATCH];
    __shared__ unsigned char crossing[DT_BATCH + 1];
    __shared__ int nitems, nslow;
    __shared__ unsigned deferred[DT_DEFER];   // pairs with a cell crossing, kept for one dense pass
    __shared__ int ndeferred;
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const long tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int start = offsets[tile * TILE_CELLS];
    const int end = offsets[(tile + 1) * TILE_CELLS];
    if (end <= start) return;
    const int tid = threadIdx.x;
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
    // One extra trip after the last batch only flushes the deferred list, so that the (large)
    // general-path code exists once in the kernel.
    for (int b0 = start; b0 < end + DT_BATCH; b0 += DT_BATCH) {
        const bool last = b0 >= end;
        const int nb = last ? 0 : min(DT_BATCH, end - b0);
        __syncthreads();   // previous round's readers are done (and the zero fill on round 0)
        // ---- stage the batch (coalesced) and key every particle by its stencil frame ----
        for (int a = tid; a < nb; a += DT_THREADS) {
            const int ip = b0 + a;
            const ParticleState p{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
            stage[0][a] = p.x; stage[1][a] = p.y; stage[2][a] = p.z; stage[3][a] = p.w;
            stage[4][a] = p.ux; stage[5][a] = p.uy; stage[6][a] = p.uz;
            int key;
            if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
                int bi, bj, bk;
                const bool cross = esirkepov_frame_cross<O>(p, g, dt, relative_time, bi, bj, bk);
                const int li = bi - o0, lj = bj - o1, lk = bk - o2;
                const bool in = li >= 0 && lj >= 0 && lk >= 0 && li + O + 3 <= N && lj + O + 3 <= N && lk + O + 3 <= N;
                key = in ? (li | (lj << 8) | (lk << 16)) : -1;
                crossing[a] = cross ? 1 : 0;
            } else {
                DirectShapes<O> s;
                direct_shapes<O>(p, g, q, relative_time, s);
                const int lo_i = min(s.jn, s.jc) - o0, lo_j = min(s.kn, s.kc) - o1, lo_k = min(s.ln, s.lc) - o2;
                const int hi_i = max(s.jn, s.jc) - o0 + O, hi_j = max(s.kn, s.kc) - o1 + O, hi_k = max(s.ln, s.lc) - o2 + O;
                key = (lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < N && hi_j < N && hi_k < N) ? 0 : -1;
            }
            if (key < 0) { sq.push(ip); key = -2 - a; }   // straggler: a key no neighbour shares
            keys[a] = key;
        }
        if (tid == 0) { nitems = 0; nslow = 0; keys[nb] = -1; }
        __syncthreads();
        // Nobody appends to the deferred list between the barrier above and the one below, so this
        // snapshot is the same in every thread (the flush decision further down must be uniform).
        const int nd0 = ndeferred;
        // ---- work items: runs of equal frames are cut into pairs (+ one single if odd) ----
        for (int a = tid; a < nb; a += DT_THREADS) {
            const int key = keys[a];
            if (key < 0) continue;
            if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
                int c = 0;
                for (int b = a; b > 0 && keys[b - 1] == key; --b) ++c;
                if ((c & 1) == 0) {
                    const bool paired = keys[a + 1] == key;
                    const bool slow = crossing[a] || (paired && crossing[a + 1]);
                    const int e = a | (paired ? PAIRED : 0);
                    // pairs without a cell crossing fill the list from the front (fast path),
                    // the others from the back (general path)
                    if (slow) items[DT_BATCH - 1 - atomicAdd(&nslow, 1)] = e;
                    else items[atomicAdd(&nitems, 1)] = e;
                }
            } else {
                items[atomicAdd(&nitems, 1)] = a;
            }
        }
        __syncthreads();
        // chunk c covers items {round*64*IPC + lane*IPC + s}: neighbouring lanes are IPC items apart,
        // i.e. one cell apart at the nominal 2*IPC particles per cell -> consecutive LDS addresses
        constexpr int IPC = 4;
        if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
            const int nfast = nitems;
            const int nfchunks = ((nfast + 64 * IPC - 1) / (64 * IPC)) * IPC;
            for (int c = wave; c < nfchunks; c += DT_THREADS / 64) {       // pairs that stay in their cell
                const int it = (c / IPC) * (64 * IPC) + lane * IPC + (c % IPC);
                if (it >= nfast) continue;
                const int e = items[it];
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
            // Pairs with a cell crossing take the general path, whose cost per wave does not depend
            // on how many lanes are active.  A handful per batch (thermal plasma: ~3 %) would cost
            // every batch a full pass, so they are deferred (as global particle indices) and run in
            // dense passes: when the list would overflow (hot / relativistic plasma: every batch)
            // and once at the end of the tile, spread over the 8 waves.
            const int nsl = last ? 0 : nslow;
            if (last || nd0 + nsl > DT_DEFER) {   // block-uniform
                const int total = nd0;
                for (int it = lane * (DT_THREADS / 64) + wave; it < total; it += DT_THREADS) {
                    const unsigned e = deferred[it];
                    const int ip = (int)(e & 0x7fffffffu);
                    const bool paired = (e & 0x80000000u) != 0;
                    const int ip2 = paired ? ip + 1 : ip;
                    const ParticleState p1{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
                    const ParticleState p2{px[ip2], py[ip2], pz[ip2], pw[ip2], pux[ip2], puy[ip2], puz[ip2]};
                    EsirkepovShapes<O> s1, s2;
                    esirkepov_shapes<O>(p1, g, q, dt, relative_time, s1);
                    esirkepov_shapes<O>(p2, g, q, dt, relative_time, s2);
                    LdsSink<M> sink(lds, s1.bi - o0, s1.bj - o1, s1.bk - o2);
                    esirkepov_accumulate_pair<O>(s1, s2, /*null2=*/!paired, g, dt, sink);
                }
                __syncthreads();
                if (tid == 0) ndeferred = 0;
                __syncthreads();
            }
            if (tid < nsl) {
                const int e = items[DT_BATCH - 1 - tid];
                deferred[atomicAdd(&ndeferred, 1)] =
                    (unsigned)(b0 + (e & (PAIRED - 1))) | ((e & PAIRED) ? 0x80000000u : 0u);
            }
            for (int it = tid + DT_THREADS; it < nsl; it += DT_THREADS) {   // nsl can reach DT_BATCH
                const int e = items[DT_BATCH - 1 - it];
                deferred[atomicAdd(&ndeferred, 1)] =
                    (unsigned)(b0 + (e & (PAIRED - 1))) | ((e & PAIRED) ? 0x80000000u : 0u);
            }
        } else {
            if (last) break;
            const int total = nitems;
            const int nchunks = ((total + 64 * IPC - 1) / (64 * IPC)) * IPC;
            for (int c = wave; c < nchunks; c += DT_THREADS / 64) {
                const int it = (c / IPC) * (64 * IPC) + lane * IPC + (c % IPC);
                if (it >= total) continue;
                const int a = items[it];
                const ParticleState p1{stage[0][a], stage[1][a], stage[2][a], stage[3][a],
                                       stage[4][a], stage[5][a], stage[6][a]};
                DirectShapes<O> s;
                direct_shapes<O>(p1, g, q, relative_time, s);
                LdsSink<M> sink(lds, -o0, -o1, -o2);
                direct_accumulate<O>(s, sink);
            }
        }
    }
    __syncthreads();
    // write-back: one global atomic per non-zero LDS point (tiles overlap on their halos)
    const DevF* Jc[3] = {&Jx, &Jy, &Jz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const DevF& J = *Jc[c];
        for (int a = tid; a < NPTS; a += DT_THREADS) {
            const double v = lds[c * NPTS + a];
            if (v != 0.0) {
                const int i = o0 + a % N, j = o1 + (a / N) % N, k = o2 + a / (N * N);
                if (i >= J.lo0 && i < J.lo0 + J.n0 && j >= J.lo1 && j < J.lo1 + J.n1 && k >= J.lo2 &&
                    k < J.lo2 + J.n2)
                    atomic_add_f64(J.p + J.off(i, j, k), v);
            }
        }
    }
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
    return ws && ws->sorted_valid && ws->sorted_x == p->x && ws->sorted_np == p->np;
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
