// LDS-tile current deposition for gfx950.
//
// The reference's own shared-memory deposition (doDepositionSharedShapeN,
// Source/Particles/Deposition/CurrentDeposition.H:453-616) handles only the direct scheme,
// one component per pass.  Here one workgroup owns one tile of 8x8x8 cells of the
// tile-major cell sort (wxa_sort_particles_by_cell), keeps all three J components of the
// tile plus its stencil halo (and one point of drift margin) in LDS as fp64
// (3 x 15 x 248 x 8 B = 89 KB), accumulates with ds_add_f64, and writes each non-zero LDS
// point back to HBM with one global fp64 atomic.  Two kernels:
//
//  * deposit_tile_rows_kernel (both algorithms): work items straight from the cell counts of the sort -- lane (r, c) of
//    a chunk takes pair r of cell 32 b + c, pairs beyond a cell's fourth come from a small tail table -- two particles
//    merged in registers per lane, the three components as three register-lean phases, crossing particles through a
//    wide-frame single body.  No staging, no per-particle keying, no barrier in the particle loop.  See the comment at the
//    kernel.
//  * deposit_stragglers_kernel: global atomics for what cannot go through the tile.
//
// Particles whose stencil leaves the LDS tile (drift of more than a cell since the last sort, particles outside the
// domain before the periodic wrap) are queued and deposited with global atomics by a second kernel, so correctness
// never depends on the sort being fresh.  DESIGN.md section 3 has the measurements behind each step.
#include "deposit_body.hpp"
#include "gather_body.hpp"
#include "heavy_tiles.hpp"
#include "workspace.hpp"

#include <stdlib.h>
#include <algorithm>
#include <type_traits>


namespace wxa {

constexpr int TS = WXA_TILE;          // tile edge in cells (workspace.hpp)
constexpr int TILE_CELLS = TS * TS * TS;

// A workgroup owns one tile of the sort, TS x TS x TSZ cells with TSZ = TS.  (Half tiles, TSZ = 4, fit two workgroups
// per CU and were slower: 8.6 ms against 5.9, profiles/round4/README.md.)
template <int M, int TSZ>
struct TileDims {
    static constexpr int LO = -2 - M;            // first LDS point relative to the tile's first cell
    static constexpr int N = TS + 5 + 2 * M;     // points per direction (x, y)
    static constexpr int NZ = TSZ + 5 + 2 * M;   // planes
    // Row stride 16 and plane stride 8 (mod 16): the LDS bank (8-byte banks, 16 per ds_add_f64 step) of point (i,j,k)
    // is (i + 8 k) mod 16 -- for the frame bases of the cells of the sort order (i fastest, then the parity of k, see
    // cell_of in particles.hip) that is the cell's index in that order mod 16, so ANY 16 consecutive cells of the
    // order, aligned or not, start on 16 different banks.
    static constexpr int NS = 16;
    static constexpr int PS = N * NS + 8;
    static constexpr int NPTS = NZ * PS;         // doubles per component (incl. padding)
};

// ACC = double: ds_add_f64, the parity build (1e-10 gate).  ACC = float: ds_add_f32 -- the throughput variant
// BASELINE.json's north_star sketches; every value is still computed in fp64 and rounded once when it enters the tile,
// the tile's sums carry fp32 round-off (the reference's own single-precision tolerance is 2e-6,
// Examples/analysis_default_regression.py:18).
template <int M, int TSZ, class ACC = double>
struct LdsSink {
    using TD = TileDims<M, TSZ>;
    ACC* base;   // LDS address of slot 0 of component 0: every deposit is base + a compile-time offset
#ifdef WXA_LDS_CHECK
    ACC* origin;
    __device__ __forceinline__ LdsSink(ACC* lds, int oi, int oj, int ok)
        : base(lds + oi + TD::NS * oj + TD::PS * ok), origin(lds) {}
#else
    __device__ __forceinline__ LdsSink(ACC* lds, int oi, int oj, int ok)
        : base(lds + oi + TD::NS * oj + TD::PS * ok) {}
#endif
    __device__ __forceinline__ void add(int c, int i, int j, int k, double v) {
#ifdef WXA_LDS_CHECK
        {
            const long at = (base - origin) + (c * TD::NPTS + i + TD::NS * j + TD::PS * k);
            if (at < 0 || at >= 3 * TD::NPTS)
                printf("LDS deposit out of the tile: at %ld (c %d i %d j %d k %d, base %ld)\n", at, c, i, j, k, (long)(base - origin));
        }
#endif
        if constexpr (sizeof(ACC) == 8) atomic_add_f64(base + (c * TD::NPTS + i + TD::NS * j + TD::PS * k), v);
        else unsafeAtomicAdd(base + (c * TD::NPTS + i + TD::NS * j + TD::PS * k), (float)v);
    }
    __device__ __forceinline__ void add_abs(int c, int gi, int gj, int gk, double v) { add(c, gi, gj, gk, v); }
    __device__ __forceinline__ LdsSink shifted(int dj, int dk) const {
        LdsSink s2 = *this;
        s2.base += TD::NS * dj + TD::PS * dk;
        return s2;
    }
};

// Phase clocks of the tile kernel (opt-in: WXA_DEPOSIT_PROFILE=1 python -m warpx_amd.build --force;
// read with scripts/deposit_profile.py).  Thread 0 of every workgroup accumulates the cycles
// between marks; a barrier before the mark makes the interval the workgroup's, not wave 0's.
#ifdef WXA_DEPOSIT_PROFILE
__device__ unsigned long long wxa_dep_prof[16];
__device__ unsigned long long wxa_dep_prof_bins[32][4];
// accumulated in registers of thread 0, added to the global counters once per workgroup (a global
// atomic per mark would serialise on the 16 addresses and distort the very thing measured)
#define DPROF_INIT                                  \
    long long prof_t = clock64();                   \
    const long long prof_t0 = prof_t;               \
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

struct JTriple {
    DevF x, y, z;
};

struct TileGeom {
    int nt[3];        // tiles per direction
    int cell_lo[3];   // global index of the brick's first cell
};

constexpr int NBANK = 16;         // LDS banks (8-byte wide) seen by one step of a ds_add_f64

// Particles whose stencil leaves the LDS tile (stale sort, particles outside the domain before
// the periodic wrap) are queued and deposited by deposit_stragglers_kernel with global atomics:
// keeping that path out of the tile kernel saves registers and instruction cache.
struct StragglerQueue {
    int* __restrict__ idx;
    unsigned* __restrict__ count;
    unsigned* __restrict__ next = nullptr;   // the next launch's counter, zeroed by this one (wxa::flip_counter)
    __device__ __forceinline__ void push(int ip) const { idx[atomicAdd(count, 1u)] = ip; }
};

__device__ __forceinline__ int frame_key(int li, int lj, int lk) { return li | (lj << 4) | (lk << 8); }

// ---- Esirkepov on LDS tiles, work items from the cell counts ----------------------------------------------------------
// The cell sort already says where every cell's particles are (offsets[]), so the work items -- (cell, r) = the cell's
// particles (2 r, 2 r + 1) -- follow from the cell counts alone, without looking at a particle:
//   * the first four pairs of every cell need no table: chunk b of 64 items is the block of 16 consecutive cells
//     16 b .. 16 b + 15, lane (r, c) = (lane / 16, lane % 16) takes pair r of cell 16 b + c.  The 16 lanes that a step of a
//     ds_add_f64 serves sit in 16 consecutive cells of the sort order = 16 different LDS banks (TileDims), never on one
//     address, and the wave's loads cover the block's particles contiguously (1 KB per array);
//   * pairs beyond the fourth (cells with more than 8 particles: ~15 % of the pairs of a thermal plasma at 8 per cell)
//     are listed in a tail table, row-major (all cells' pair 4, then pair 5, ...), built from one ballot per row and one
//     wave scan of the 64 (row, cell-wave) counts -- dense chunks again, processed by the same loop.
// Phases: A cell counts, row masks, zero fill | B scan + tail table | C the chunks: two particles per lane (14 loads in
// flight), merged when both stay in their cell and share the frame, phased pair body; what cannot be merged or crosses
// a cell goes to the deferred list | D deferred particles through the wide single body, one lane per (component,
// particle) | E write-back.  Cells with more than 8 + 2 RT particles hand the rest to the deferred list as well.
template <int NT_, int WPE_, class ACC_ = double, int ALGO_ = WXA_DEPOSIT_ESIRKEPOV, int WL_ = 0>
struct RowsCfg {
    // WL (wide frames in the loop): every particle is deposited inside the chunk loop through the wide-frame single body
    // (O + 2 slots per direction: correct whether it crosses a cell or not), nothing is merged, nothing deferred for
    // crossing.  For a plasma that STREAMS through the grid -- a boosted-frame run: every particle moves c dt / dz = 0.76
    // cells per step against the boost, three in four cross a cell -- where the fast body's premise (1-2 % cross) fails:
    // the deferred list (2048 per tile) overflowed into the global-atomics pass and the boosted wakefield deck at
    // 256 x 256 x 512 x 8 per cell spent 660 ms per launch there (profiles/round5/README.md).  300 LDS atomics per
    // particle instead of 72: the price of a crossing particle, paid on the tile.  Chosen per workspace
    // (wxa_workspace_set_streaming_plasma); the default kernel is untouched.
    static constexpr int WL = WL_;
    static constexpr int NT = NT_, WPE = WPE_;
    static constexpr int ALGO = ALGO_;   // WXA_DEPOSIT_DIRECT: the same work items, the lane's two particles on one frame per component
    using ACC = ACC_;   // accumulator type of the LDS tile
    // cells per block of the direct part: a chunk is BW consecutive cells x 64 / BW pairs.  32 consecutive cells of the
    // sort order start on 32 different 4-byte banks (i + 16 j + 24 k) and -- lanes l and l + 32 being two pairs of one
    // cell -- a ds_add_f64 never meets the same addresses in consecutive 16-lane steps (11 cycles per wave instruction
    // instead of 8 with blocks of 16 cells x 4 pairs, profiles/round3/lds_atomic_microbench.txt)
    static constexpr int BW = 32;
};

template <int O, int M, class CFG>
__global__ void __launch_bounds__(CFG::NT) WXA_WAVES_PER_SIMD(CFG::WPE)
deposit_tile_rows_kernel(JTriple J3, const double* __restrict__ px, const double* __restrict__ py,
                         const double* __restrict__ pz, const double* __restrict__ pw,
                         const double* __restrict__ pux, const double* __restrict__ puy,
                         const double* __restrict__ puz, const int* __restrict__ offsets, Geom g, TileGeom tg, double q,
                         EsirkepovStep es, double relative_time, StragglerQueue sq, HeavyUnits hu) {
    constexpr int NT = CFG::NT, TSZ = TS;
    using TD = TileDims<M, TSZ>;
    constexpr int N = TD::N, NZ = TD::NZ, NPTS = TD::NPTS, PS = TD::PS;
    constexpr int CELLS = TILE_CELLS;              // cells of a unit (= a tile of the sort)
    constexpr int CW = CELLS / 64;                 // cell-waves (waves that hold a cell per lane in phases A and B)
    using ACC = typename CFG::ACC;
    constexpr int BW = CFG::BW, RPC = 64 / BW;     // cells per block, pairs (rows) of a cell per chunk
    constexpr int NB = (CELLS / BW) * (4 / RPC);   // chunks of the direct part (the first four pairs of every cell)
    constexpr int RT = 64 / CW;                    // rows of the tail table (pairs 4 .. 3 + RT): RT CW = 64 counts = one wave scan
    constexpr int RMAX = 4 + RT;
    // WL (a streaming plasma; lanes l and l + 32 of a chunk share their deposits where they sit in one cell): a work item
    // of the tail table and of the excess is a cell's pairs (2 m, 2 m + 1) for the lanes l and l + 32, like the direct part's
    // -- 32 items per chunk, RT / 2 table rows.  Otherwise an item is one pair, 64 per chunk (consecutive lanes in
    // consecutive cells: the layout the uniform plasma's bank arithmetic is made for).
    constexpr int TW = CFG::WL != 0 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV ? 2 : 1;   // pairs per tail / excess item
    // ... and a tile of a streaming plasma with few particles (the plasma's edge, the cavity behind the pulse: a twelfth of
    // the boosted wakefield's particles, a third of its workgroups' time until round 6) lists ALL its pairs in the table
    // and skips the direct part, whose 32 chunks would each run the whole body for the two or three lanes in 64 that hold
    // a particle: SPARSE_MAX particles fill at most SPARSE_MAX / 4 + CELLS = 1024 entries, the table's capacity.
    constexpr bool CAN_BE_SPARSE = TW == 2;
    constexpr int SPARSE_MAX = 2048;
    constexpr int TCAP = CELLS * 2;                // tail capacity (8 ppc: 0.66 tail items per cell on average)
    static_assert(!CAN_BE_SPARSE || SPARSE_MAX / 4 + CELLS <= TCAP, "a sparse tile's pairs fit the table");
    constexpr int DEFER = 2048;
    static_assert(NT >= CELLS && RT >= 8, "one lane per cell");
    __shared__ ACC lds[3 * NPTS];
    __shared__ unsigned long long masks[RT][CW];
    __shared__ int cstart[CELLS + 1];
    __shared__ unsigned short table[TCAP];
    // particles with a cell crossing or without a partner (wide body), bucketed by the LDS bank of their wide frame:
    // phase D takes lane l's particle from bucket l % 16, so the 16 lanes of a ds_add_f64 step sit on 16 different banks
    // like the lanes of phase C (the single list it replaces cost 2-3 LDS cycles per step in conflicts: the list
    // pass was 16 % of the kernel for 3 % of the particles)
    // Odd orders only: at an even order the stencil frame follows the NEAREST node, the particles of a sort cell fall into
    // eight frames and most of them have no partner -- the lone list (half the deferred entries) overflowed into the
    // global-atomics pass (order 2, 256^3 x 8 ppc: 37 ms per launch against 8.4 with one list), and two particles of a
    // lane rarely share the direct deposition's frame (9.5 ms against 8.4 one after the other).
    // SNG: a second deferred list for the particles that stay in their cell but cannot be merged with their lane partner
    // (another stencil frame: one of the two has left the sort cell since the last sort).  Phase D runs them through a
    // one-component body on their own fast frame, (O+1)^2 O = 48 atomics per component, instead of the crossing
    // particles' wide body with (O+2)^2 (O+1) = 100.  Half of phase D's entries are of this kind at a sort interval of 3.
    constexpr bool SNG = CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && (O & 1) != 0;
    // the same on the direct deposition: the lane's two particles on one frame, the ones without a partner deferred
    constexpr bool DPM = CFG::ALGO == WXA_DEPOSIT_DIRECT && (O & 1) != 0;
    constexpr int NBKT = SNG ? 2 * NBANK : NBANK;   // buckets: by bank for the crossing particles, then by bank for the lone ones
    __shared__ unsigned deferred[DEFER];
    __shared__ int ndef[NBKT];
    __shared__ int nitems;
    // Cells beyond the tables (more than 2 RMAX particles: a wake's density spike holds thousands): their excess pairs are
    // more chunks of the same loop -- item j of the excess is found through the running sums xs[] of the cells' excess
    // pairs (binary search in LDS) -- so that they deposit on the tile like everybody else.  (Until round 5 one lane
    // deferred them particle by particle, by index, and what the deferred list could not take went to the global-atomics
    // pass: the boosted wakefield deck at 8 per cell spent 600 ms per launch there, thousands of particles of one cell
    // adding to the same hundred addresses of J in L2.)  Tiles without such a cell: one LDS flag read.
    __shared__ int xs[CELLS + 1];
    __shared__ int any_excess;
    // ... and a tile whose pairs beyond the fourth do not fit the tail table (a compressed plasma: 16 particles per cell
    // on average fill its 1024 entries) keeps no tail at all: every pair beyond a cell's fourth is an excess pair.  (What
    // the table could not take used to go to the deferred list and on to the global-atomics pass: 12 % of the particles
    // of the boosted wakefield deck, 41 ms per launch, profiles/round5/README.md.)
    __shared__ int tail_off;
    // ... and their data, for the first DKEEP entries of every bucket: phase C has the particle in registers when it
    // defers it; fetched again by index in phase D each one costs seven cache lines from HBM (the tile's lines have left
    // the L2 by then: FETCH_SIZE 1.57 x the particle data, phase D 12 % of the kernel for 3 % of the particles)
    constexpr int DKEEP = (sizeof(ACC) == 8 ? 48 : 16) / (SNG ? 2 : 1);   // 16 x 48 x 56 B = 42 KB next to the 89 KB tile
    __shared__ double dkeep[7][NBKT * DKEEP];
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == 0 && tid == 0 && sq.next) *sq.next = 0u;
    DPROF_INIT
    int unit_u = 0, unit_k = 1;   // this workgroup takes the chunks ch = unit_u (mod unit_k) of its tile
    long unit;
    // (a tile with far more particles than the others is shared by several workgroups: heavy_tiles.hpp)
    if (!heavy_unit_of(hu, blockIdx.x, ntiles, unit, unit_u, unit_k)) return;
    const long tile = unit;
    const long ucell0 = tile * TILE_CELLS;
    const int start = offsets[ucell0];
    const int end = offsets[ucell0 + CELLS];
    if (end <= start) return;
    const bool sparse = CAN_BE_SPARSE && end - start <= SPARSE_MAX;   // uniform
    const int first_row_pair = sparse ? 0 : 4;                        // the table's rows start at this pair of a cell
    const int RTU = (RMAX - first_row_pair) / TW;                     // table rows in use
    constexpr int DCAP = DEFER / NBKT;
    auto defer = [&](const int ip, const int bank) {   // phase B: by index only (the particle has not been loaded)
        const int n = atomicAdd(&ndef[bank], 1);
        if (n < DCAP) deferred[bank * DCAP + n] = (unsigned)ip | 0x80000000u;
        else sq.push(ip);
    };
    // phase B's overflow cases (a cell with more than 2 RMAX particles, a full tail table) never pass through the chunk
    // loop.  Esirkepov: the deferred list (phase D's wide body).  Direct deposition: the straggler kernel -- phase D is
    // Esirkepov's body (until round 3 these particles went there whatever the algorithm: one cell in a million at 8 per
    // cell, seen first at 256^3).
    auto defer_unloaded = [&](const int ip, const int bank) {
        if (unit_u != 0) return;   // a tile shared by several workgroups: what is deferred by index is unit 0's
        if constexpr (CFG::ALGO == WXA_DEPOSIT_DIRECT) sq.push(ip);
        else defer(ip, bank);
    };
    // (false: the bucket is full, nothing was done)
    auto try_defer_particle = [&](const int ip, const int bank, const ParticleState& pp) {
        const int n = atomicAdd(&ndef[bank], 1);
        if (n >= DCAP) return false;
        const bool keep = n < DKEEP;
        deferred[bank * DCAP + n] = (unsigned)ip | (keep ? 0u : 0x80000000u);
        if (keep) {
            const int at = bank * DKEEP + n;
            dkeep[0][at] = pp.x; dkeep[1][at] = pp.y; dkeep[2][at] = pp.z; dkeep[3][at] = pp.w;
            dkeep[4][at] = pp.ux; dkeep[5][at] = pp.uy; dkeep[6][at] = pp.uz;
        }
        return true;
    };
    auto defer_particle = [&](const int ip, const int bank, const ParticleState& pp) {
        if (!try_defer_particle(ip, bank, pp)) sq.push(ip);
    };
    constexpr int WAVES = NT / 64;
    // ---- A: cell counts, row masks; zero fill
    if (tid == 0) { nitems = 0; any_excess = 0; tail_off = 0; }
    if (tid < NBKT) ndef[tid] = 0;
    int my_s = 0, my_n = 0, my_pairs = 0;
    unsigned long long my_mask[RT];   // wave-uniform: tail row r of this cell-wave
    // the two offsets of the lane's cell are requested first and the tile is zeroed while they travel
    int off_lo = 0, off_hi = 0;
    if (tid < CELLS) { off_lo = offsets[ucell0 + tid]; off_hi = offsets[ucell0 + tid + 1]; }
    for (int a = tid; a < 3 * NPTS; a += NT) lds[a] = (ACC)0;
    if (tid < CELLS) {
        my_s = off_lo;
        my_n = off_hi - my_s;
        cstart[tid] = my_s;
        if (tid == CELLS - 1) cstart[CELLS] = my_s + my_n;
        my_pairs = min((my_n + 1) >> 1, RMAX);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            my_mask[r] = __ballot(r < RTU && my_pairs > first_row_pair + TW * r);
            if (lane == 0) masks[r][wave] = my_mask[r];
        }
    }
    __syncthreads();
    DPROF(0);
    // ---- B: scan of the 64 (row, cell-wave) counts, item table
    if (tid < CELLS) {
        // scan order of the 64 (row, cell-wave) counts: row-major
        const int cnt = __popcll(masks[lane / CW][lane % CW]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int excl = incl - cnt;
        const int total = __shfl(incl, 63);
        const bool no_tail = total > TCAP;   // the same in every wave
        if (tid == 0) { nitems = no_tail ? 0 : min(total, TCAP); tail_off = no_tail; }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int base = __shfl(excl, r * CW + wave);   // first item of (tail row r, this cell-wave)
            if (r < RTU && my_pairs > first_row_pair + TW * r && !no_tail) {
                const int at = base + __popcll(my_mask[r] & lt);
                if (at < TCAP) table[at] = (unsigned short)(tid | ((first_row_pair + TW * r) << 9));
                else {   // any bucket is correct; the cell's place in the sort order is the bank of a particle that stayed
                    for (int e = 2 * (first_row_pair + TW * r); e < min(2 * (first_row_pair + TW * r) + 2 * TW, my_n); ++e)
                        defer_unloaded(my_s + e, tid & (NBANK - 1));
                }
            }
        }
        // beyond the table's rows (> 2 RMAX particles in a cell): chunks of their own
        if (my_n > 2 * (no_tail ? 4 : RMAX)) any_excess = 1;
    }
    __syncthreads();
    int excess_pairs = 0;
    const int rmax_t = __builtin_amdgcn_readfirstlane(tail_off) ? 4 : RMAX;   // pairs of a cell that the direct chunks and the tail table cover
    if (any_excess) {   // uniform
        if (tid < CELLS) xs[tid] = (((max(0, my_n - 2 * rmax_t) + 1) >> 1) + TW - 1) / TW;   // items of the cell's excess
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int c = 0; c < CELLS; ++c) { const int v = xs[c]; xs[c] = acc; acc += v; }
            xs[CELLS] = acc;
        }
        __syncthreads();
        excess_pairs = __builtin_amdgcn_readfirstlane(xs[CELLS]);
    }
    DPROF(1);
    const int ti = (int)(tile % tg.nt[0]);
    const int tj = (int)((tile / tg.nt[0]) % tg.nt[1]);
    const int tk = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
    const int o0 = tg.cell_lo[0] + ti * TS + TD::LO;
    const int o1 = tg.cell_lo[1] + tj * TS + TD::LO;
    const int o2 = tg.cell_lo[2] + tk * TS + TD::LO;
    // ---- C: the chunks (NB blocks of 16 cells x 4 pairs, then the tail table)
    const int T = min(nitems, TCAP);
    constexpr int TPC = 64 / TW;   // tail / excess items per chunk
#ifdef WXA_DEPOSIT_PROFILE
    const long long prof_l0 = clock64();
#endif
    const int CH0 = unit_u + unit_k * wave, CHS = unit_k * WAVES;   // the chunks of this unit, wave by wave
    const int NBr = sparse ? 0 : NB;   // chunks of the direct part that run
    const int nregular = NBr + ((T + TPC - 1) / TPC);
    const int nchunks = nregular + ((excess_pairs + TPC - 1) / TPC);   // (excess_pairs: the excess' items)
    // the lane's work item of chunk ch: its two particles (an empty lane reads the tile's first particle)
    auto item_of = [&](const int ch, int& ia, int& ib, bool& va, bool& vb) {
        int c, r;
        if (ch < NBr) {
            c = BW * (ch / (4 / RPC)) + (lane % BW); r = RPC * (ch % (4 / RPC)) + lane / BW; va = true;
        } else if (ch < nregular) {
            const int I = (ch - NBr) * TPC + (lane & (TPC - 1));
            va = I < T;
            const unsigned ent = va ? table[I] : 0u;
            c = (int)(ent & 511u); r = (int)(ent >> 9) + lane / TPC;
        } else {   // pair I of the cells' excess: the cell whose running sum holds it, xs[c] <= I < xs[c + 1]
            // Lane (r, s) = (lane / 16, lane % 16) of excess chunk j takes pair 4 j + r of stream s, the streams being
            // sixteen equal ranges of the excess pairs: the 16 lanes that a step of a ds_add_f64 serves sit in sixteen
            // different parts of the tile (consecutive pairs -- one cell, one frame, 64 lanes on the same addresses --
            // made the excess chunks of a dense tile several times slower than its other chunks), and every stream
            // still reads four neighbouring pairs = 64 contiguous bytes per array.
            // (WL: items of two pairs, lane (r, s) = (lane % 32 / 16, lane % 16) takes the first pair of item 2 j + r of stream s,
            // lane + 32 the second)
            const int nexc = (excess_pairs + TPC - 1) / TPC;
            const int I = (lane & 15) * ((4 / TW) * nexc) + (4 / TW) * (ch - nregular) + ((lane & (TPC - 1)) >> 4);
            va = I < excess_pairs;
            int lo = 0, hi = CELLS;
            const int J = va ? I : 0;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (xs[mid] <= J) lo = mid; else hi = mid;
            }
            r = rmax_t + TW * (J - xs[lo]) + lane / TPC;
            c = lo;
        }
        const int s0 = cstart[c], n0 = cstart[c + 1] - s0;
        va = va && 2 * r < n0;
        ia = va ? s0 + 2 * r : start;
        vb = va && 2 * r + 1 < n0;
        ib = vb ? ia + 1 : ia;
    };
    for (int ch = CH0; ch < nchunks; ch += CHS) {   // wave-uniform
        int ia, ib;
        bool va, vb;
        item_of(ch, ia, ib, va, vb);
        // all fourteen loads in flight together (an empty lane reads the tile's first particle)
#ifdef WXA_DEPOSIT_PROFILE
        const long long prof_c0 = clock64();
#endif
        const ParticleState pa{px[ia], py[ia], pz[ia], pw[ia], pux[ia], puy[ia], puz[ia]};
        const ParticleState pb{px[ib], py[ib], pz[ib], pw[ib], pux[ib], puy[ib], puz[ib]};
#ifdef WXA_DEPOSIT_PROFILE   // wave 0 of every workgroup: cycles from the loads' issue to their arrival, and of the whole chunk
        long long prof_c1 = 0;
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prof_c1 = clock64();
        }
#endif
        if constexpr (DPM) {
            // doDepositionShapeN (CurrentDeposition.H:48-249) on the tile, the lane's two particles on one frame per
            // component (direct_pair_component): they share it when their nodal stencils coincide (two particles of a
            // sort cell).  A second particle on another frame goes to the deferred list (phase D: one lane per
            // component and particle on the particle's own stencils); a stencil that leaves the tile to the
            // global-atomics pass.  The range check is the frame's, one point wider than the particle's own.
            DirectShapes<O> da, db;
            direct_shapes<O>(pa, g, q, relative_time, da);
            direct_shapes<O>(pb, g, q, relative_time, db);
            auto inside = [&](const DirectShapes<O>& d) {
                const int li = d.jn - 1 - o0, lj = d.kn - 1 - o1, lk = d.ln - 1 - o2;
                return li >= 0 && lj >= 0 && lk >= 0 && li + O + 1 < N && lj + O + 1 < N && lk + O + 1 < NZ;
            };
            bool a_on = va && inside(da), b_on = vb && inside(db);
            if (va && !a_on) sq.push(ia);
            if (vb && !b_on) sq.push(ib);
            if (a_on && b_on && (da.jn != db.jn || da.kn != db.kn || da.ln != db.ln)) {
                defer_particle(ib, ((db.jn - o0) + 8 * (db.ln - o2)) & (NBANK - 1), pb);
                b_on = false;
            }
            if (a_on || b_on) {
                if (!a_on) da.wqx = da.wqy = da.wqz = 0.0;
                if (!b_on) db.wqx = db.wqy = db.wqz = 0.0;
                const int fi = (a_on ? da.jn : db.jn) - o0, fj = (a_on ? da.kn : db.kn) - o1, fk = (a_on ? da.ln : db.ln) - o2;
                LdsSink<M, TSZ, ACC> sjx(lds, fi - 1, fj, fk), sjy(lds, fi, fj - 1, fk), sjz(lds, fi, fj, fk - 1);
                direct_pair_component<O, 0>(da, db, sjx);
                direct_pair_component<O, 1>(da, db, sjy);
                direct_pair_component<O, 2>(da, db, sjz);
            }
            continue;
        }
        if constexpr (CFG::ALGO == WXA_DEPOSIT_DIRECT && !DPM) {
            // doDepositionShapeN (CurrentDeposition.H:48-249) on the tile: the lane's two particles one after the other,
            // each component on its own frame (jx: cell-centred in x, nodal in y and z; ...), products in the
            // reference's order; a stencil that leaves the tile goes to the global-atomics pass
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                if (!(h ? vb : va)) continue;
                const ParticleState pp = h ? pb : pa;
                DirectShapes<O> ds;
                direct_shapes<O>(pp, g, q, relative_time, ds);
                const int li = min(ds.jn, ds.jc) - o0, hi = max(ds.jn, ds.jc) + O - o0;
                const int lj = min(ds.kn, ds.kc) - o1, hj = max(ds.kn, ds.kc) + O - o1;
                const int lk = min(ds.ln, ds.lc) - o2, hk = max(ds.ln, ds.lc) + O - o2;
                if (!(li >= 0 && lj >= 0 && lk >= 0 && hi < N && hj < N && hk < NZ)) {
                    sq.push(h ? ib : ia);
                    continue;
                }
                LdsSink<M, TSZ, ACC> sjx(lds, ds.jc - o0, ds.kn - o1, ds.ln - o2);
                LdsSink<M, TSZ, ACC> sjy(lds, ds.jn - o0, ds.kc - o1, ds.ln - o2);
                LdsSink<M, TSZ, ACC> sjz(lds, ds.jn - o0, ds.kn - o1, ds.lc - o2);
#pragma unroll
                for (int iz = 0; iz <= O; iz++)
#pragma unroll
                    for (int iy = 0; iy <= O; iy++)
#pragma unroll
                        for (int ix = 0; ix <= O; ix++) {
                            sjx.add(0, ix, iy, iz, ds.sxc[ix] * ds.syn[iy] * ds.szn[iz] * ds.wqx);
                            sjy.add(1, ix, iy, iz, ds.sxn[ix] * ds.syc[iy] * ds.szn[iz] * ds.wqy);
                            sjz.add(2, ix, iy, iz, ds.sxn[ix] * ds.syn[iy] * ds.szc[iz] * ds.wqz);
                        }
            }
            continue;
        }
        EsirkepovCoords c1 = esirkepov_coords(pa, g, es), c2 = esirkepov_coords(pb, g, es);
        if constexpr (CFG::WL != 0 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV) {
            // The chunks of the cells' excess pairs (a density spike: 10^3 .. 10^5 particles in a cell) are 64 lanes on one or
            // two cells: the lanes that share a frame -- hu.wave_sum_min or more of them -- sum every value over the wave
            // (wave_sum_f64, VALU only) and one lane adds it, instead of 64 lanes adding to one LDS address one after
            // the other.  Wave-uniform control flow throughout: every lane computes, with weight 0 where it has no part.
            const bool crowded = ch >= nregular;   // wave-uniform
            WideFrame<O> fa = esirkepov_wide_frame<O>(c1, g), fb = esirkepov_wide_frame<O>(c2, g);
            auto fits = [&](const WideFrame<O>& f) {
                const int wi = f.b[0] - o0, wj = f.b[1] - o1, wk = f.b[2] - o2;
                return wi >= 0 && wj >= 0 && wk >= 0 && wi + O + 2 <= N && wj + O + 2 <= N && wk + O + 2 <= NZ;
            };
            bool on_a = va && fits(fa), on_b = vb && fits(fb);
            if (va && !on_a) sq.push(ia);   // the frame leaves the tile: the global-atomics pass
            if (vb && !on_b) sq.push(ib);
            if (__ballot(on_a || on_b) == 0ull) continue;   // (a fifth of the boosted wakefield's chunks: cells ahead of the plasma)
            double wqa = q * pa.w, wqb = q * pb.w;
            if (crowded) {
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    const EsirkepovCoords cc = h ? c2 : c1;
                    const WideFrame<O> f = h ? fb : fa;
                    bool on = h ? on_b : on_a;
                    const double wq = h ? wqb : wqa;
                    const int key = on ? ((f.b[0] - o0) | ((f.b[1] - o1) << 8) | ((f.b[2] - o2) << 16)) : -1;
                    unsigned long long rest = __ballot(on);
                    while (rest) {   // one trip per large group; the first small one ends the search
                        const int k0 = __shfl(key, __ffsll((long long)rest) - 1);
                        const bool same = on && key == k0;
                        const unsigned long long group = __ballot(same);
                        rest &= ~group;
                        if (__popcll(group) < hu.wave_sum_min) break;
                        WaveSumSink<LdsSink<M, TSZ, ACC>> sink(LdsSink<M, TSZ, ACC>(lds, k0 & 255, (k0 >> 8) & 255, k0 >> 16), lane == 63);
                        const double wqs = same ? wq : 0.0;
                        // (its own copy of the coordinates: the weights are not to be kept for the lane's own pass below)
                        EsirkepovCoords cr = cc;
                        WXA_OPAQUE_F64(cr.x_new); WXA_OPAQUE_F64(cr.y_new); WXA_OPAQUE_F64(cr.z_new);
                        WXA_OPAQUE_F64(cr.x_old); WXA_OPAQUE_F64(cr.y_old); WXA_OPAQUE_F64(cr.z_old);
                        esirkepov_single_wide<O, 0>(cr, f, wqs, es, sink);
                        esirkepov_single_wide<O, 1>(cr, f, wqs, es, sink);
                        esirkepov_single_wide<O, 2>(cr, f, wqs, es, sink);
                        on = on && !same;
                    }
                    if (h) on_b = on; else on_a = on;
                }
            }
            // The lane's two particles come from one cell; where their wide frames start at the same point -- the stream
            // runs one way, so they mostly do -- every point takes the sum of the two values: 150 LDS atomics per particle
            // instead of 300 (the LDS array is what this kernel waits for: 73 % busy, the fp64 VALU 14 %,
            // profiles/round5/README.md).  Without a partner on its frame, particle a goes through the same body alone.
            // (A body that is wide along z only -- 4 x 4 x 5 points, 184 values instead of 300, for particles that stay in
            // their cell in x and y -- was measured and dropped: inside the wake the laser shakes every particle across
            // x, the deferred list overflowed with the crossing ones (61 -> 472 ms per launch for BASELINE config 5), and
            // chosen wave by wave it changed nothing (60.9 ms): sessions t and u.)
            // A particle that stays in its cell along d fills slots 0 .. O of its frame there (both offsets 0); it sits as well
            // on slots 1 .. O + 1 of the frame that starts one point lower -- the frame of a neighbour that crosses downwards.
            // So particles of one cell whose frames start one point apart along some directions are brought onto ONE frame
            // where the higher one can be lowered (of a stream that moves 0.9 cells per step nine in ten cross and one does
            // not: without this nearly every wave holds a lane whose two particles need a pass each, and no wave's lane
            // pairs agree).  A frame lowered to another's start fits the tile where that one does.
            auto can_lower = [](const WideFrame<O>& f, const int d) { return f.sn[d] == 0 && f.so[d] == 0; };
            auto lower = [](WideFrame<O>& f, const int d) { f.sn[d] = f.so[d] = 1; f.b[d] -= 1; };
            if (!on_a && on_b) {   // the second particle alone takes the first one's place (one pass instead of two)
                c1 = c2; fa = fb; wqa = wqb;
                on_a = true; on_b = false;
            }
            bool merged = on_a && on_b;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int diff = fa.b[d] - fb.b[d];
                if (merged && diff == 1 && can_lower(fa, d)) lower(fa, d);
                else if (merged && diff == -1 && can_lower(fb, d)) lower(fb, d);
                else if (diff != 0) merged = false;
            }
            // Lanes l and l + 32 hold two pairs of one cell: the same between the two lanes' frames.  A frame's key: its start
            // on the tile, a byte per direction, and in the top byte the directions along which it can still be lowered; a
            // lane without a particle joins the other's frame.
            constexpr int EMPTY = -1, BROKEN = -2;
            int low[3] = {0, 0, 0};   // this lane's frame comes down along d to the pair's
            auto unify = [&](const int mine, const int theirs) {
                if (theirs < 0) return mine;
                if (mine < 0) return theirs;
                int out = mine & 0xffffff;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int mb = (mine >> (8 * d)) & 255, tb = (theirs >> (8 * d)) & 255;
                    const int mc = (mine >> (24 + d)) & 1, tc = (theirs >> (24 + d)) & 1;
                    if (mb == tb + 1 && mc) { out -= 1 << (8 * d); low[d] = 1; }
                    else if (mb + 1 == tb && tc) { }   // (the other lane comes down)
                    else if (mb != tb) return BROKEN;
                }
                return out;
            };
            int mine = EMPTY;
            if (on_a) {
                int cs = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) cs |= (can_lower(fa, d) && (!merged || can_lower(fb, d))) << d;
                mine = (fa.b[0] - o0) | ((fa.b[1] - o1) << 8) | ((fa.b[2] - o2) << 16) | (cs << 24);
            }
            const int theirs = partner32(mine);
            const int pairkey = unify(mine, theirs);
            const bool shared = mine >= 0 && theirs >= 0 && pairkey != BROKEN;   // two particles' lanes on one frame
            const bool pairs = (O & 1) && __ballot(pairkey == BROKEN) == 0ull;    // wave-uniform: which way of adding
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (low[d] && on_a) {
                    lower(fa, d);
                    if (merged) lower(fb, d);
                }
            // (the frames packed, and unpacked again for every component behind a fence: registers)
            PackedWideFrame qa = pack_wide_frame<O>(fa), qb = pack_wide_frame<O>(fb);
            auto fence = [&]() {
                WXA_OPAQUE_I32(qa.jn[0]); WXA_OPAQUE_I32(qa.jn[1]); WXA_OPAQUE_I32(qa.jn[2]); WXA_OPAQUE_I32(qa.bits);
                WXA_OPAQUE_I32(qb.jn[0]); WXA_OPAQUE_I32(qb.jn[1]); WXA_OPAQUE_I32(qb.jn[2]); WXA_OPAQUE_I32(qb.bits);
                WXA_OPAQUE_F64(c1.x_new); WXA_OPAQUE_F64(c1.y_new); WXA_OPAQUE_F64(c1.z_new);
                WXA_OPAQUE_F64(c1.x_old); WXA_OPAQUE_F64(c1.y_old); WXA_OPAQUE_F64(c1.z_old);
                WXA_OPAQUE_F64(c2.x_new); WXA_OPAQUE_F64(c2.y_new); WXA_OPAQUE_F64(c2.z_new);
                WXA_OPAQUE_F64(c2.x_old); WXA_OPAQUE_F64(c2.y_old); WXA_OPAQUE_F64(c2.z_old);
            };
            {
                // Where the two lanes deposit on one frame their values are summed: every lane of the wave runs the body,
                // a lane without a particle on the tile with zero weights (the exchange between the lanes is a wave operation).
                const double wq1 = on_a ? wqa : 0.0, wq2 = on_a && merged ? wqb : 0.0;
                const bool adds = on_a && !(shared && lane >= 32);
#ifdef WXA_DEPOSIT_PROFILE   // every wave's chunks: how many take which body, how many lanes hold a particle, how many share a frame
                {
                    const unsigned long long m_on = __ballot(on_a), m_sh = __ballot(shared), m_2 = __ballot(on_b && !merged);
                    if (lane == 0) {
                        atomicAdd(&wxa_dep_prof[10], 1ull);
                        atomicAdd(&wxa_dep_prof[11], (unsigned long long)pairs);
                        atomicAdd(&wxa_dep_prof[12], 0ull);
                        atomicAdd(&wxa_dep_prof[13], (unsigned long long)__popcll(m_2));
                        atomicAdd(&wxa_dep_prof[14], (unsigned long long)__popcll(m_sh));
                        atomicAdd(&wxa_dep_prof[15], (unsigned long long)__popcll(m_on));
                    }
                }
#endif
                // Every lane pair of the wave on one frame (a lane without a particle joins its partner's with zero weights):
                // the two lanes share the summing AND the adding, PairScatterSink -- half the LDS-atomic instructions.
                // Otherwise the lower lane of a pair that shares a frame adds the pair's sums (PairSumSink).
                if (pairs) {
                    const bool active = pairkey >= 0;
                    const int kk = active ? pairkey : 0, up = lane >> 5;
                    auto component = [&](auto comp) {
                        constexpr int COMP = decltype(comp)::value;
                        fence();
                        const WideFrame<O> f1 = unpack_wide_frame<O>(qa, g), f2 = unpack_wide_frame<O>(qb, g);
                        PairScatterSink<LdsSink<M, TSZ, ACC>> sink(
                            LdsSink<M, TSZ, ACC>(lds, (kk & 255) + (COMP == 0 ? up : 0), ((kk >> 8) & 255) + (COMP == 1 ? up : 0),
                                                 ((kk >> 16) & 255) + (COMP == 2 ? up : 0)), active);
                        esirkepov_pair_wide<O, COMP>(c1, f1, wq1, c2, f2, wq2, es, sink);
                    };
                    component(std::integral_constant<int, 0>{});
                    component(std::integral_constant<int, 1>{});
                    component(std::integral_constant<int, 2>{});
                } else {
                    auto component = [&](auto comp) {
                        fence();
                        const WideFrame<O> f1 = unpack_wide_frame<O>(qa, g), f2 = unpack_wide_frame<O>(qb, g);
                        PairSumSink<LdsSink<M, TSZ, ACC>> sink(LdsSink<M, TSZ, ACC>(lds, f1.b[0] - o0, f1.b[1] - o1, f1.b[2] - o2), shared, adds);
                        esirkepov_pair_wide<O, decltype(comp)::value>(c1, f1, wq1, c2, f2, wq2, es, sink);
                    };
                    component(std::integral_constant<int, 0>{});
                    component(std::integral_constant<int, 1>{});
                    component(std::integral_constant<int, 2>{});
                }
            }
            // A second particle that cannot share the first one's frame (2 % of the boosted wakefield's: the laser shakes
            // them across x both ways) goes to the deferred list -- phase D's dense waves, one lane per component and
            // particle -- as long as its bucket has room: a pass of its own through the wide body costs the whole wave 300
            // LDS atomics for the four lanes in 64 that need it (a third of that deck's chunks hold such a lane).
            bool second = on_b && !merged;
            if (second) second = !try_defer_particle(ib, ((fb.b[0] - o0) + 8 * (fb.b[2] - o2)) & (NBANK - 1), pb);
            if (second) {
                auto component = [&](auto comp) {
                    fence();
                    const WideFrame<O> f2 = unpack_wide_frame<O>(qb, g);
                    LdsSink<M, TSZ, ACC> sink(lds, f2.b[0] - o0, f2.b[1] - o1, f2.b[2] - o2);
                    esirkepov_single_wide<O, decltype(comp)::value>(c2, f2, wqb, es, sink);
                };
                component(std::integral_constant<int, 0>{});
                component(std::integral_constant<int, 1>{});
                component(std::integral_constant<int, 2>{});
            }
            continue;
        }
        double wq1 = q * pa.w, wq2 = 0.0;
        const double wqb = q * pb.w;
        int ai, aj, ak, bi, bj, bk;
        const bool crossa = esirkepov_frame_cross<O>(c1, g, ai, aj, ak);
        const bool crossb = esirkepov_frame_cross<O>(c2, g, bi, bj, bk);
        auto inside = [&](int li, int lj, int lk) {
            return li >= 0 && lj >= 0 && lk >= 0 && li + O + 3 <= N && lj + O + 3 <= N && lk + O + 3 <= NZ;
        };
        const bool ina = inside(ai - o0, aj - o1, ak - o2), inb = inside(bi - o0, bj - o1, bk - o2);
        const int ka = frame_key(ai - o0, aj - o1, ak - o2), kb = frame_key(bi - o0, bj - o1, bk - o2);
        const int sa = !va ? 3 : !ina ? 2 : crossa ? 1 : 0;   // 0: fast, 1: general path on the tile, 2: straggler, 3: none
        const int sb = !vb ? 3 : !inb ? 2 : crossb ? 1 : 0;
        // LDS bank (8-byte banks, 16 per step) of the first point of the particle's wide frame: i + 8 k (TileDims)
        auto wide_bank = [&](const EsirkepovCoords& cc) {
            const WideFrame<O> f = esirkepov_wide_frame<O>(cc, g);
            return ((f.b[0] - o0) + 8 * (f.b[2] - o2)) & (NBANK - 1);
        };
        if (sa == 2) sq.push(ia);
        if (sb == 2) sq.push(ib);
        if (sa == 1) defer_particle(ia, wide_bank(c1), pa);
        if (sb == 1) defer_particle(ib, wide_bank(c2), pb);
        if constexpr ((O & 1) == 0) {
            // Even orders: the frame follows the nearest node, so the particles of a sort cell sit on eight frames and a
            // lane's two particles share one only by chance -- and a pair on the union frame would cost more atomics
            // ((O + 2)^2 (O + 1) per component) than two particles on their own ((O + 1)^2 O each).  So the lane deposits its
            // two particles one after the other, each on its own frame: 2 x 54 LDS atomics at order 2, nothing deferred
            // but the crossing particles.  (Through the pair body with the partner deferred: 8.4 ms per launch at
            // 256^3 x 8 ppc, more than order 3's 5.9.)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                if ((h ? sb : sa) != 0) continue;
                const EsirkepovCoords cc = h ? c2 : c1;
                LdsSink<M, TSZ, ACC> sink(lds, (h ? bi : ai) - o0, (h ? bj : aj) - o1, (h ? bk : ak) - o2);
                const double wq = h ? wqb : wq1;
                esirkepov_single_fast<O, 0>(cc, wq, es, sink);
                esirkepov_single_fast<O, 1>(cc, wq, es, sink);
                esirkepov_single_fast<O, 2>(cc, wq, es, sink);
            }
            continue;
        }
        int key = -1;
        if (sa == 0) {
            key = ka;
            if (sb == 0 && kb == ka) {
                wq2 = wqb;                               // merged with its neighbour
            } else {
                // another frame than its lane partner: deferred, alone, to the lone list (its fast frame's bank)
                if (sb == 0) defer_particle(ib, NBANK + (((kb & 15) + 8 * (kb >> 8)) & (NBANK - 1)), pb);
                c2 = c1;                                 // empty partner
            }
        } else if (sb == 0) {
            key = kb; c1 = c2; wq1 = wqb;                // the second particle alone
        }
        if (key >= 0) {
            LdsSink<M, TSZ, ACC> sink(lds, key & 15, (key >> 4) & 15, key >> 8);
            esirkepov_pair_phased<O, false>(c1, c2, wq1, wq2, es, sink);
        }
#ifdef WXA_DEPOSIT_PROFILE
        if (wave == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the chunk's atomics have left the queue
            const long long prof_c2 = clock64();
            if (tid == 0) { DCOUNT(6, prof_c1 - prof_c0); DCOUNT(7, prof_c2 - prof_c1); DCOUNT(8, 1); }
        }
#endif
    }
#ifdef WXA_DEPOSIT_PROFILE
    if (tid == 0) DCOUNT(9, clock64() - prof_l0);   // wave 0's own time in the loop; the rest of phase 2 is its wait at the barrier
#endif
    __syncthreads();
    DPROF(2);
    {   // ---- D: the deferred particles through the wide body, one lane per (component, particle).  A chunk of 64
        //      lanes belongs to ONE component (until round 3 the lanes of a wave sat in different component branches; the
        //      fp32 build then deposited the stencil's far corner wrongly for some lanes -- see below for what that was).
        // a chunk = 4 rows of the 16 buckets: lane l takes entry 4 (chunk) + l / 16 of bucket l % 16
        // chunks of the crossing particles (buckets 0 .. 15), then -- SNG -- of the lone ones (16 .. 31), numbered through
        auto rows_of = [&](const int first_bucket) {
            int n = min(ndef[first_bucket + (lane & (NBANK - 1))], DCAP);
#pragma unroll
            for (int d = 1; d < NBANK; d <<= 1) n = max(n, __shfl_xor(n, d));
            return (n + 3) >> 2;
        };
        const int dch0 = rows_of(0), dch1 = SNG ? rows_of(NBANK) : 0;
        // One loop per (kind, component), both compile-time constants inside it.  As one loop with the component a
        // wave-uniform scalar -- `if (comp == 0) ... else if (comp == 1) ... else ...` -- the gfx950 build merges the last
        // ds_add of the three bodies into one shared block whose address register is an implicit-def on the edge from
        // comp >= 2 (seen in the ISA of the fp32 order-2 build, round 4: the last deposit of every jz body went to a
        // stale address; a far corner of 1e-5 for the wide body, a 4 % error and a memory fault for the lone
        // particles' frame).  With nothing to choose between there is nothing to merge.  The chunks stay numbered
        // through all six loops -- chunk n belongs to wave n % WAVES -- so the waves' shares are what they were.
        auto pass = [&](auto comp_c, auto lone_c, const int base, const int dch) {
            constexpr int COMP = decltype(comp_c)::value;
            constexpr bool LONE = decltype(lone_c)::value;
            int first = (wave - base) % WAVES;
            if (first < 0) first += WAVES;
            for (int cc = first; cc < dch; cc += WAVES) {
                const int row = cc * 4 + (lane >> 4);
                const int bkt = (LONE ? NBANK : 0) + (lane & (NBANK - 1));
                if (row < min(ndef[bkt], DCAP)) {
                    const unsigned ent = deferred[bkt * DCAP + row];
                    const int ip = (int)(ent & 0x7fffffffu);
                    ParticleState p1;
                    if (ent & 0x80000000u) {
                        p1 = ParticleState{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
                    } else {
                        const int at = bkt * DKEEP + row;
                        p1 = ParticleState{dkeep[0][at], dkeep[1][at], dkeep[2][at], dkeep[3][at], dkeep[4][at],
                                           dkeep[5][at], dkeep[6][at]};
                    }
                    if constexpr (DPM) {
                        DirectShapes<O> ds;
                        direct_shapes<O>(p1, g, q, relative_time, ds);
                        LdsSink<M, TSZ, ACC> sink(lds, (COMP == 0 ? ds.jc : ds.jn) - o0, (COMP == 1 ? ds.kc : ds.kn) - o1,
                                                  (COMP == 2 ? ds.lc : ds.ln) - o2);
                        direct_single_component<O, COMP>(ds, sink);
                        continue;
                    }
                    const EsirkepovCoords c1 = esirkepov_coords(p1, g, es);
                    const double wq = q * p1.w;
                    if constexpr (LONE) {
                        int fi, fj, fk;
                        esirkepov_frame_cross<O>(c1, g, fi, fj, fk);
                        LdsSink<M, TSZ, ACC> sink(lds, fi - o0, fj - o1, fk - o2);
                        esirkepov_single_fast<O, COMP>(c1, wq, es, sink);
                    } else {
                        const WideFrame<O> f = esirkepov_wide_frame<O>(c1, g);
                        // The particles that phase B deferred by index (beyond the 24th of a cell, a full tail table) were
                        // never seen by the chunk loop and its range check: a frame that leaves the tile (the sort's age,
                        // a particle moving a cell per step) goes to the global-atomics pass, once (the pass of component
                        // 0 queues it), instead of into the lists that lie behind the tile in LDS.  (Found in round 5 by
                        // the boosted wakefield deck at 8 per cell: a density spike, a corrupted deferred list, a memory
                        // fault; test_deposit_current_lds_tiles_crowded_cells[drift].)
                        const int wi = f.b[0] - o0, wj = f.b[1] - o1, wk = f.b[2] - o2;
                        if (!(wi >= 0 && wj >= 0 && wk >= 0 && wi + O + 2 <= N && wj + O + 2 <= N && wk + O + 2 <= NZ)) {
                            if constexpr (COMP == 0) sq.push(ip);
                            continue;
                        }
                        LdsSink<M, TSZ, ACC> sink(lds, wi, wj, wk);
                        esirkepov_single_wide<O, COMP>(c1, f, wq, es, sink);
                    }
                }
            }
        };
        using std::integral_constant;
        pass(integral_constant<int, 0>{}, integral_constant<bool, false>{}, 0, dch0);
        pass(integral_constant<int, 1>{}, integral_constant<bool, false>{}, dch0, dch0);
        pass(integral_constant<int, 2>{}, integral_constant<bool, false>{}, 2 * dch0, dch0);
        if constexpr (SNG) {
            pass(integral_constant<int, 0>{}, integral_constant<bool, true>{}, 3 * dch0, dch1);
            pass(integral_constant<int, 1>{}, integral_constant<bool, true>{}, 3 * dch0 + dch1, dch1);
            pass(integral_constant<int, 2>{}, integral_constant<bool, true>{}, 3 * dch0 + 2 * dch1, dch1);
        }
    }
    __syncthreads();
    DPROF(3);
    {
        // E: a lane owns one (i, j) column of one component and walks it along k -- all NZ LDS reads of the column in
        // flight at once, one address computation per column instead of one 64-bit index decomposition per point.  (Point
        // by point -- ds_read, s_waitcnt lgkmcnt(0), compare, ~20 integer instructions of address arithmetic, atomic,
        // fifteen times in a row per lane -- the write-back was 7 % of the kernel.)
        const JTriple* jt = &J3;
        constexpr int NSC = TD::NS, COLS = N * NSC;   // columns of a component, the padding points of a row included
        for (int col = tid; col < 3 * COLS; col += NT) {
            const int c = col / COLS, ij = col - c * COLS;
            const int li = ij % NSC, lj = ij / NSC;
            ACC* src = lds + c * NPTS + ij;
            double v[NZ];
#pragma unroll
            for (int k = 0; k < NZ; ++k) v[k] = (double)src[k * PS];
            const DevF J = c == 0 ? jt->x : c == 1 ? jt->y : jt->z;
            const int i = o0 + li, j = o1 + lj;
            if (li >= N || i < J.lo0 || i >= J.lo0 + J.n0 || j < J.lo1 || j >= J.lo1 + J.n1) continue;
            double* dst = J.p + ((long)(i - J.lo0) + (long)(j - J.lo1) * J.js);
#pragma unroll
            for (int k = 0; k < NZ; ++k) {
                const int kk = o2 + k - J.lo2;
                if (v[k] != 0.0 && kk >= 0 && kk < J.n2) atomic_add_f64(dst + (long)kk * J.ks, v[k]);
            }
        }
    }
    DPROF(4);
    DPROF_FINISH
#ifdef WXA_DEPOSIT_PROFILE   // the workgroups' lengths by the size of their share: cycles, count, the longest (bins of log2 particles)
    if (tid == 0) {
        const int share = max(1, (end - start) / unit_k);
        const int bin = 31 - __clz(share);
        const unsigned long long cyc = (unsigned long long)(clock64() - prof_t0);
        atomicAdd(&wxa_dep_prof_bins[bin][0], cyc);
        atomicAdd(&wxa_dep_prof_bins[bin][1], 1ull);
        atomicMax(&wxa_dep_prof_bins[bin][2], cyc);
        atomicAdd(&wxa_dep_prof_bins[bin][3], (unsigned long long)share);
    }
#endif
}

template <int O, int ALGO>
__global__ void __launch_bounds__(256)
deposit_stragglers_kernel(const double* __restrict__ px, const double* __restrict__ py,
                          const double* __restrict__ pz, const double* __restrict__ pw,
                          const double* __restrict__ pux, const double* __restrict__ puy,
                          const double* __restrict__ puz, const int* __restrict__ idx,
                          const unsigned* __restrict__ count, DevF Jx, DevF Jy, DevF Jz, Geom g, double q,
                          EsirkepovStep es, double relative_time) {
    const unsigned n = *count;
    GlobalSink gs = make_global_sink(Jx, Jy, Jz);
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int ip = idx[t];
        const ParticleState p{px[ip], py[ip], pz[ip], pw[ip], pux[ip], puy[ip], puz[ip]};
        if constexpr (ALGO == WXA_DEPOSIT_ESIRKEPOV) {
            EsirkepovShapes<O> s;
            esirkepov_shapes<O>(p, g, q, es, s);
            gs.bi = s.bi; gs.bj = s.bj; gs.bk = s.bk;
            esirkepov_accumulate<O>(s, es, gs);
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

// One extra LDS point per side beyond the exact stencil reach (M = 1): particles may have drifted up
// to one cell since the last sort (sort_intervals > 1) before they have to take the
// straggler path (global atomics, ~7 ns per particle).
constexpr int MARGIN = 1;

template <int O, class CFG>
static wxa_status launch_rows(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* geom,
                               double q, double dt, double relative_time, wxa_workspace* ws, hipStream_t st) {
    TileGeom tg;
    for (int d = 0; d < 3; ++d) {
        tg.nt[d] = (ws->sort_nc[d] + TS - 1) / TS;
        tg.cell_lo[d] = ws->sort_cell_lo[d];
    }
    const long nunits = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const Geom g = make_geom(*geom);
    const int* offsets = (const int*)ws->offsets.p;
    wxa_status rc;
    // tiles with far more particles than the others are shared by several workgroups (heavy_tiles.hpp)
    HeavyUnits hu;
    long extra_groups = 0;
    if ((rc = plan_heavy_tiles(ws, offsets, nunits, (long)p->np, hu, extra_groups, st)) != WXA_OK) return rc;
    const dim3 grid((unsigned)(xcd_grid_size(nunits) + extra_groups)), block(CFG::NT);
    if ((rc = ws->stragglers.reserve(sizeof(int) * (size_t)p->np + 64)) != WXA_OK) return rc;
    StragglerQueue sq{(int*)ws->stragglers.p, nullptr, nullptr};
    unsigned *cnt_now = nullptr, *cnt_next = nullptr;
    if ((rc = flip_counter(ws, 0, ws->deposit_flips, st, cnt_now, cnt_next)) != WXA_OK) return rc;   // words 0, 1 of ws->counters
    sq.count = cnt_now; sq.next = cnt_next;
    const DevF jx = make_devf(J[0]), jy = make_devf(J[1]), jz = make_devf(J[2]);
    const EsirkepovStep es = make_esirkepov_step(g, dt, relative_time);
    hipLaunchKernelGGL((deposit_tile_rows_kernel<O, MARGIN, CFG>), grid, block, 0, st, JTriple{jx, jy, jz}, p->x, p->y, p->z, p->w,
                       p->ux, p->uy, p->uz, offsets, g, tg, q, es, relative_time, sq, hu);
    hipLaunchKernelGGL((deposit_stragglers_kernel<O, CFG::ALGO>), dim3(512), dim3(256), 0, st, p->x, p->y,
                       p->z, p->w, p->ux, p->uy, p->uz, sq.idx, sq.count, jx, jy, jz, g, q, es, relative_time);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// Production configurations: whole tiles, 768 lanes = 3 waves per SIMD (168 VGPRs), chunks of 32 cells x 2 pairs.  What was
// measured on the way and removed again -- half tiles, 8 / 10 / 11 / 16 waves, hole filling, persistent tiles, dynamic
// chunks, lane pairs sharing their deposits, the next chunk's particles in registers, 16-byte loads, a fused push + deposit
// kernel, ... -- is listed with its numbers in profiles/kernel_history_rounds_1_to_4.md and profiles/round6/README.md;
// the code of those variants is in the history of this file (round 5's head), not in the product.
using RowsEsirkepov = RowsCfg<768, 3, double, WXA_DEPOSIT_ESIRKEPOV>;
using RowsEsirkepovF32 = RowsCfg<768, 3, float, WXA_DEPOSIT_ESIRKEPOV>;   // fp32 tile accumulation (ds_add_f32), opt-in per workspace
using RowsDirect = RowsCfg<768, 3, double, WXA_DEPOSIT_DIRECT>;
using RowsEsirkepovStreaming = RowsCfg<768, 3, double, WXA_DEPOSIT_ESIRKEPOV, 1>;   // WL

wxa_status deposit_current_tiled(const wxa_particle_view* p, const wxa_field_view J[3], const wxa_grid_geom* geom,
                                 double q, double dt, double relative_time, int order, int algo,
                                 wxa_workspace* ws, hipStream_t st) {
    if (algo == WXA_DEPOSIT_ESIRKEPOV) {
        if (ws->deposit_accumulator == WXA_ACC_FP32) {
            if (order == 1) return launch_rows<1, RowsEsirkepovF32>(p, J, geom, q, dt, relative_time, ws, st);
            if (order == 2) return launch_rows<2, RowsEsirkepovF32>(p, J, geom, q, dt, relative_time, ws, st);
            return launch_rows<3, RowsEsirkepovF32>(p, J, geom, q, dt, relative_time, ws, st);
        }
        if (ws->streaming_plasma) {   // wxa_workspace_set_streaming_plasma: most particles cross a cell per step
            if (order == 1) return launch_rows<1, RowsEsirkepovStreaming>(p, J, geom, q, dt, relative_time, ws, st);
            if (order == 2) return launch_rows<2, RowsEsirkepovStreaming>(p, J, geom, q, dt, relative_time, ws, st);
            return launch_rows<3, RowsEsirkepovStreaming>(p, J, geom, q, dt, relative_time, ws, st);
        }
        if (order == 1) return launch_rows<1, RowsEsirkepov>(p, J, geom, q, dt, relative_time, ws, st);
        if (order == 2) return launch_rows<2, RowsEsirkepov>(p, J, geom, q, dt, relative_time, ws, st);
        // order 4 (round 6): the tile's 15 points per direction are exactly the reach of the quartic stencil around a tile
        // (nodes c - 2 .. c + 3 of the particle's cell c, one more each way for the old position) -- no point of drift
        // margin is left, so a particle that has left its sort cell by more than the stencil's slack goes to the
        // global-atomics pass; everything else deposits on the tile like the other even order
        if (order == 4) return launch_rows<4, RowsEsirkepov>(p, J, geom, q, dt, relative_time, ws, st);
        return launch_rows<3, RowsEsirkepov>(p, J, geom, q, dt, relative_time, ws, st);
    }
    if (order == 1) return launch_rows<1, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
    if (order == 2) return launch_rows<2, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
    if (order == 4) return launch_rows<4, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
    return launch_rows<3, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
}

}  // namespace wxa

#ifdef WXA_DEPOSIT_PROFILE
extern "C" int wxa_debug_deposit_profile_bins(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(wxa::wxa_dep_prof_bins), sizeof(wxa::wxa_dep_prof_bins)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[32 * 4] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wxa::wxa_dep_prof_bins), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
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
