// LDS-tile current deposition for gfx950.
//
// The reference's own shared-memory deposition (doDepositionSharedShapeN,
// Source/Particles/Deposition/CurrentDeposition.H:453-616) handles only the direct scheme,
// one component per pass.  Here one workgroup owns one tile of 8x8x8 cells of the
// tile-major cell sort (wxa_sort_particles_by_cell), keeps all three J components of the
// tile plus its stencil halo in LDS as fp64 (3 x 13^3 x 8 B = 52.7 KB -> 3 workgroups per
// CU), accumulates with ds_add_f64, and writes each non-zero LDS point back to HBM with
// one global fp64 atomic.  Particles whose stencil leaves the LDS tile (drift since the
// last sort, domain-edge particles before the periodic wrap) take the global-atomic path,
// so correctness never depends on the sort being fresh.
//
// Lanes walk the tile's particle range in chunks (lane l handles particles
// l*chunk .. l*chunk+chunk-1): at any instant the lanes of a wave work on particles of
// different cells, which spreads the LDS atomics over different addresses instead of
// serialising the ~8 same-cell particles of a cell-sorted wave on one address.
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
    double* lds;
    int oi, oj, ok;   // local index of slot 0 (relative form) or of grid index 0 (absolute form)
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
        constexpr int N = TileDims<M>::N;
        atomic_add_f64(lds + c * TileDims<M>::NPTS + (oi + i) + N * ((oj + j) + N * (ok + k)), v);
    }
    __device__ __forceinline__ void add_abs(int c, int gi, int gj, int gk, double v) { add(c, gi, gj, gk, v); }
};

struct TileGeom {
    int nt[3];        // tiles per direction
    int cell_lo[3];   // global index of the brick's first cell
};

constexpr int DT_THREADS = 512;   // 8 waves: one workgroup per CU (LDS-limited), 2 waves per SIMD
constexpr int DT_BATCH = 512;     // particles staged in LDS per round (7 x 512 x 8 B = 28 KB)

template <int O, int ALGO, int M>
__global__ void __launch_bounds__(DT_THREADS)
deposit_tile_kernel(const double* __restrict__ px, const double* __restrict__ py,
                    const double* __restrict__ pz, const double* __restrict__ pw,
                    const double* __restrict__ pux, const double* __restrict__ puy,
                    const double* __restrict__ puz, const int* __restrict__ offsets, DevF Jx, DevF Jy,
                    DevF Jz, Geom g, TileGeom tg, double q, double dt, double relative_time) {
    constexpr int N = TileDims<M>::N;
    constexpr int NPTS = TileDims<M>::NPTS;
    __shared__ double lds[3 * NPTS];
    __shared__ double stage[7][DT_BATCH];
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
    GlobalSink gs = make_global_sink(Jx, Jy, Jz);
    // staged slot handled by this lane: a stride-8 walk, so that the 64 lanes of a wave hold
    // particles ~8 apart in the cell-sorted order (different cells at ~8 ppc) and their LDS
    // atomics fall on different addresses
    const int slot = (tid * 8) % DT_BATCH + (tid * 8) / DT_BATCH;

    for (int b0 = start; b0 < end; b0 += DT_BATCH) {
        const int nb = min(DT_BATCH, end - b0);
        __syncthreads();   // previous round's readers are done (and the zero fill on round 0)
        if (tid < nb) {    // coalesced: consecutive lanes read consecutive particles
            const int ip = b0 + tid;
            stage[0][tid] = px[ip]; stage[1][tid] = py[ip]; stage[2][tid] = pz[ip]; stage[3][tid] = pw[ip];
            stage[4][tid] = pux[ip]; stage[5][tid] = puy[ip]; stage[6][tid] = puz[ip];
        }
        __syncthreads();
        if (slot >= nb) continue;
        ParticleState p{stage[0][slot], stage[1][slot], stage[2][slot], stage[3][slot],
                        stage[4][slot], stage[5][slot], stage[6][slot]};
        if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
            EsirkepovShapes<O> s;
            esirkepov_shapes<O>(p, g, q, dt, relative_time, s);
            const int li = s.bi - o0, lj = s.bj - o1, lk = s.bk - o2;
            if (li >= 0 && lj >= 0 && lk >= 0 && li + O + 3 <= N && lj + O + 3 <= N && lk + O + 3 <= N) {
                LdsSink<M> sink{lds, li, lj, lk};
                esirkepov_accumulate<O>(s, g, dt, sink);
            } else {
                gs.bi = s.bi; gs.bj = s.bj; gs.bk = s.bk;
                esirkepov_accumulate<O>(s, g, dt, gs);
            }
        } else {
            DirectShapes<O> s;
            direct_shapes<O>(p, g, q, relative_time, s);
            const int lo_i = min(s.jn, s.jc) - o0, lo_j = min(s.kn, s.kc) - o1, lo_k = min(s.ln, s.lc) - o2;
            const int hi_i = max(s.jn, s.jc) - o0 + O, hi_j = max(s.kn, s.kc) - o1 + O, hi_k = max(s.ln, s.lc) - o2 + O;
            if (lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < N && hi_j < N && hi_k < N) {
                LdsSink<M> sink{lds, -o0, -o1, -o2};
                direct_accumulate<O>(s, sink);
            } else {
                direct_accumulate<O>(s, gs);
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
    // Esirkepov with relative_time = -dt/2 deposits at the sorted positions: no margin needed;
    // otherwise (direct at the half step, or a stale sort) keep one extra point per side.
    constexpr int M = (ALGO == WXA_DEPOSIT_ESIRKEPOV) ? 0 : 1;
    hipLaunchKernelGGL((deposit_tile_kernel<O, ALGO, M>), grid, block, 0, st, p->x, p->y, p->z, p->w, p->ux,
                       p->uy, p->uz, offsets, make_devf(J[0]), make_devf(J[1]), make_devf(J[2]), g, tg, q, dt,
                       relative_time);
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
