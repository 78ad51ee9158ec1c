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

// A workgroup owns TS x TS x TSZ cells: a whole tile of the sort (TSZ = 8) or its lower / upper half in z
// (TSZ = 4; the sort order inside a tile has k/2 as its slowest index, so a half tile is a contiguous range
// of cells and of particles).  Half tiles need 61 KB instead of 83.5 KB of LDS: two workgroups per CU.
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

// The J arrays as the kernel's FIRST parameter: offset 0 of the kernel-argument segment (WXA_LATE_KERNARG)
struct JTriple {
    DevF x, y, z;
};
#ifndef WXA_OPAQUE_UNIFORM_F64   // a wave-uniform double in an SGPR pair, opaque to the optimiser (tests/hipcpu: nothing)
#define WXA_OPAQUE_UNIFORM_F64(v) asm volatile("" : "+s"(v))
#endif
#ifndef WXA_LATE_KERNARG   // tests/hipcpu: the parameter itself
#define WXA_LATE_KERNARG(T, first_param)                                                                         \
    ([]() {                                                                                                      \
        auto p_ = (const T __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();           \
        asm volatile("" : "+s"(p_));                                                                             \
        return (const T*)p_;                                                                                     \
    }())
#endif

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

// CFG::FUSED: PhysicalParticleContainer::Evolve (PhysicalParticleContainer.cpp:1812-2095) on a tile in one kernel -- the
// chunk loop gathers E and B for its two particles from a staged tile of the six field components (the gather tile
// kernel's 6 x 11^3 doubles), pushes them, stores position and momentum, and deposits the pair from registers: the
// deposition's re-read of the particles (56 B of the 164 B per particle and step) and its exposed load latency are gone.
// LDS: 89 KB J tile + 64 KB field tile + tables = 158 KB, so the deferred particles are not kept in LDS (they are read
// back after the chunk loop's barrier) and their list is half as long.  Order 3 with the energy-conserving gather only.
struct FusedArgs {
    PV p;                                // the same arrays as px .. puz, writable
    DevF Ex, Ey, Ez, Bx, By, Bz;
    Geom gg;                             // geometry of the E / B arrays (the gather's index origin)
    double m, dt;
    StragglerQueue gq;                   // particles whose gather stencil leaves the staged tile: pushed AND deposited later
};
constexpr int FUSED_GN = TS + 3;         // staged field points per direction (gather_tile.hip, GatherTileDims<1>)
constexpr int FUSED_GLO = -1;
constexpr int FUSED_GNPTS = FUSED_GN * FUSED_GN * FUSED_GN;

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
// DBG (timing experiments only): 1 = the arithmetic without the LDS atomics, 2 = the LDS atomics without the arithmetic
template <int NT_, int TSZ_, int WPE_, int PHASED_, int DBG_ = 0, class ACC_ = double, int BW_ = 0,
          int ALGO_ = WXA_DEPOSIT_ESIRKEPOV, int COOP_ = 0, int DYN_ = 0, int FUSED_ = 0, int PUSHER_ = WXA_PUSHER_BORIS,
          int HF_ = 0, int PT_ = 0, int GIDX_ = 0, int FLUSH_ = 0, int ZF_ = 0, int SNG_ = 0, int PFD_ = 0, int TI_ = 0, int TB_ = 0, int LD16_ = 0,
          int WL_ = 0>
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
    // LD16: the lane's two particles -- neighbours in every array -- come as one 16-byte load per array (7 load instructions
    // per chunk instead of 14, each lane's bytes in one place)
    static constexpr int LD16 = LD16_;
    static constexpr int TB = TB_;   // 1: the tiles in blocks of 4 x 4 x 2 (blocked_tile, common.hpp)
    // TI: the tail table is ordered by cell (cell-wave, then row) instead of by row, and its chunks are interleaved with
    // the direct chunks in proportion, so that a cell's pairs beyond the fourth are read a few chunks after its first
    // four instead of ~35 chunks later, when their cache lines have left the L2 (a tile's particles are 230 KB, the L2
    // holds 128 KB per CU): the loop's loads alone take 2.1 ms, what 12.5 GB of traffic cost -- 8.3 GB are algorithmic
    static constexpr int TI = TI_;
    // PFD: the particles of a wave's NEXT chunk are requested in the middle of the current chunk's pair body (1: before its
    // last component, 2: before its second) instead of at the top of their own chunk: without it a wave has loads in
    // flight only while it waits for them -- the kernel without arithmetic and without atomics still takes 3.8 of its 5.9 ms
    static constexpr int PFD = PFD_;
    // SNG: a second deferred list for the particles that stay in their cell but cannot be merged with their lane partner
    // (another stencil frame: one of the two has left the sort cell since the last sort).  Phase D runs them through a
    // one-component body on their own fast frame, (O+1)^2 O = 48 atomics per component, instead of the crossing
    // particles' wide body with (O+2)^2 (O+1) = 100.  Half of phase D's entries are of this kind at a sort interval of 3.
    static constexpr int SNG = SNG_;
    static constexpr int ZF = ZF_;         // 1: zero fill behind the loads of the cell offsets (see phase A)
    static constexpr int FLUSH = FLUSH_;   // 1: the write-back by columns (see phase E)
    // GIDX: the direct chunks read their cell's first particle and count from the sort's offsets[] in global memory
    // (vmcnt) instead of from the LDS copy: an LDS read at the top of a chunk returns behind whatever the CU's waves
    // have queued on the LDS-atomic pipe (lgkmcnt counts the wave's own atomics too), and the fourteen particle loads
    // cannot be issued before it is back
    static constexpr int GIDX = GIDX_;
    // PT (persistent tiles): one workgroup per CU works through the tiles of its XCD's range (claimed through a counter per
    // XCD; other XCDs' ranges once its own is done) instead of one workgroup per tile.  With 145 KB of LDS a CU holds one
    // workgroup, so every tile paid a workgroup launch after the previous one had retired, i.e. after its last global
    // atomic had come back; here the flush's atomics drain behind the next tile's phases A-C, and the flush leaves the
    // tile zeroed (read + clear by the same lane), so only a workgroup's first tile needs the zero fill of phase A.
    static constexpr int PT = PT_;
    // HF (hole filling): an empty slot of the direct part -- lane (r, c) of a cell with fewer than r + 1 pairs -- takes a
    // pair beyond the fourth of another cell of the SAME bank class (cell index mod BW: the same lane position, the same
    // LDS banks), matched by rank inside the class; only what the class's holes cannot take goes to the tail table.  At
    // 8 per cell (Poisson) the direct part has 10.7 % of its slots empty and the tail holds 0.69 pairs per cell: 38 chunks
    // per tile become ~35, i.e. three rounds of the workgroup's 12 waves instead of three and a fourth with two waves
    // working (wave 0 idles 22 % of phase C, profiles/round3/README.md), and a cell's late pairs are read while their
    // cache lines are still near (the tail table re-read them ~40 chunks later: 1.5 x the algorithmic HBM traffic).
    static constexpr int HF = HF_;
    static constexpr int FUSED = FUSED_, PUSHER = PUSHER_;   // gather + push inside the chunk loop (FusedArgs)
    // DYN: the chunks of phase C are handed out through an LDS counter instead of chunk = wave + k WAVES: the SIMD's
    // issue arbiter favours its oldest waves, so with equal static shares the youngest waves of every SIMD finish last and
    // the others wait at the barrier (profile build: wave 0 idles 22 % of the phase, profiles/round3/README.md)
    static constexpr int DYN = DYN_;
    // COOP: lanes l and l + 32 of a chunk (pairs r and r + 2 of one cell) share their deposits, each lane issues half the
    // LDS atomics (esirkepov_pair_phased_coop; odd orders, fp64 tiles)
    static constexpr int COOP = COOP_;
    static constexpr int NT = NT_, TSZ = TSZ_, WPE = WPE_, PHASED = PHASED_, DBG = DBG_;
    static constexpr int ALGO = ALGO_;   // WXA_DEPOSIT_DIRECT: the same work items, every particle on its own (no pairs)
    using ACC = ACC_;   // accumulator type of the LDS tile
    // cells per block of the direct part = lanes that one step of the LDS atomic serves (16 for ds_add_f64, 32 for
    // ds_add_f32): a chunk is BW consecutive cells x 64 / BW pairs, and the first four pairs of a block take
    // 4 BW / 64 chunks.  32 consecutive cells of the sort order start on 32 different 4-byte banks (i + 16 j + 24 k).
    static constexpr int BW = BW_ ? BW_ : (sizeof(ACC_) == 8 ? 16 : 32);
};

struct NullSink {   // DBG = 1: keeps every deposited value alive without touching the LDS
    double acc = 0.0;
    __device__ __forceinline__ void add(int, int, int, int, double v) { acc += v; }
};

template <int O, int M, class CFG>
__global__ void __launch_bounds__(CFG::NT) WXA_WAVES_PER_SIMD(CFG::WPE)
deposit_tile_rows_kernel(JTriple J3, const double* __restrict__ px_, const double* __restrict__ py_,
                         const double* __restrict__ pz_, const double* __restrict__ pw_,
                         const double* __restrict__ pux_, const double* __restrict__ puy_,
                         const double* __restrict__ puz_, const int* __restrict__ offsets, Geom g_, TileGeom tg, double q_,
                         EsirkepovStep es_, double relative_time_, StragglerQueue sq, FusedArgs fa,
                         unsigned* __restrict__ tile_ctr, HeavyUnits hu) {
    constexpr int NT = CFG::NT, TSZ = CFG::TSZ;
    constexpr bool FUSED = CFG::FUSED != 0;
    static_assert(!FUSED || (O == 3 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && CFG::TSZ == TS && sizeof(typename CFG::ACC) == 8),
                  "the fused kernel: order 3, Esirkepov, whole tiles, fp64 tiles");
    const double *px, *py, *pz, *pw, *pux, *puy, *puz;
    if constexpr (FUSED) {   // every access goes through the writable views (the const restrict parameters stay unused)
        px = fa.p.x; py = fa.p.y; pz = fa.p.z; pw = fa.p.w; pux = fa.p.ux; puy = fa.p.uy; puz = fa.p.uz;
    } else {
        px = px_; py = py_; pz = pz_; pw = pw_; pux = pux_; puy = puy_; puz = puz_;
    }
    using TD = TileDims<M, TSZ>;
    constexpr int N = TD::N, NZ = TD::NZ, NPTS = TD::NPTS, PS = TD::PS;
    constexpr int SUB = TS / TSZ;
    constexpr int CELLS = TILE_CELLS / SUB;        // cells of this unit
    constexpr int CW = CELLS / 64;                 // cell-waves (waves that hold a cell per lane in phases A and B)
    using ACC = typename CFG::ACC;
    constexpr int BW = CFG::BW, RPC = 64 / BW;     // cells per block, pairs (rows) of a cell per chunk
    constexpr int NB = (CELLS / BW) * (4 / RPC);   // chunks of the direct part (the first four pairs of every cell)
    constexpr int RT = 64 / CW;                    // rows of the tail table (pairs 4 .. 3 + RT): RT CW = 64 counts = one wave scan
    constexpr int RMAX = 4 + RT;
    constexpr int TCAP = CELLS * 2;                // tail capacity (8 ppc: 0.66 tail items per cell on average)
    constexpr int DEFER = TSZ == 8 && !FUSED ? 2048 : 1024;
    static_assert(NT >= CELLS && RT >= 8, "one lane per cell");
    __shared__ ACC lds[3 * NPTS];
    __shared__ unsigned long long masks[RT][CW];
    __shared__ int cstart[CELLS + 1];
    __shared__ unsigned short table[TCAP];
    constexpr bool HF = CFG::HF != 0;
    constexpr int NCLS = BW, MEM = CELLS / BW;          // bank classes of the cells (cell mod BW) and cells per class
    static_assert(!HF || (MEM >= 2 && MEM <= 64 && (MEM & (MEM - 1)) == 0 && !FUSED), "hole filling: a class is a power-of-two lane segment");
    constexpr unsigned short NOFILL = 0xFFFFu;
    __shared__ unsigned short slot[HF ? 4 : 1][HF ? CELLS : 1];   // hole (r, c) of the direct part -> (cell | row << 9) that fills it
    __shared__ unsigned short fill[HF ? 4 * CELLS : 1];           // per class: its pairs beyond the fourth, by rank
    // particles with a cell crossing or without a partner (wide body), bucketed by the LDS bank of their wide frame:
    // phase D takes lane l's particle from bucket l % 16, so the 16 lanes of a ds_add_f64 step sit on 16 different banks
    // like the lanes of phase C (the single list it replaces cost 2-3 LDS cycles per step in conflicts: the list
    // pass was 16 % of the kernel for 3 % of the particles)
    // Odd orders only: at an even order the stencil frame follows the NEAREST node, the particles of a sort cell fall into
    // eight frames and most of them have no partner -- the lone list (half the deferred entries) overflowed into the
    // global-atomics pass (order 2, 256^3 x 8 ppc: 37 ms per launch against 8.4 with one list), and two particles of a
    // lane rarely share the direct deposition's frame (9.5 ms against 8.4 one after the other).
    constexpr bool SNG = CFG::SNG != 0 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && !FUSED && (O & 1) != 0;
    // the same switch on the direct deposition: the lane's two particles on one frame, the ones without a partner deferred
    constexpr bool DPM = CFG::SNG != 0 && CFG::ALGO == WXA_DEPOSIT_DIRECT && !FUSED && (O & 1) != 0;
    constexpr int NBKT = SNG ? 2 * NBANK : NBANK;   // buckets: by bank for the crossing particles, then by bank for the lone ones
    __shared__ unsigned deferred[DEFER];
    __shared__ int ndef[NBKT];
    __shared__ int nitems;
    __shared__ int next_chunk;
    // Cells beyond the tables (more than 2 RMAX particles: a wake's density spike holds thousands): their excess pairs are
    // more chunks of the same loop -- item j of the excess is found through the running sums xs[] of the cells' excess
    // pairs (binary search in LDS) -- so that they deposit on the tile like everybody else.  (Until round 5 one lane
    // deferred them particle by particle, by index, and what the deferred list could not take went to the global-atomics
    // pass: the boosted wakefield deck at 8 per cell spent 600 ms per launch there, thousands of particles of one cell
    // adding to the same hundred addresses of J in L2.)  Tiles without such a cell: one LDS flag read.
    constexpr bool XCH = !FUSED && CFG::HF == 0 && CFG::TI == 0 && CFG::COOP == 0;
    __shared__ int xs[XCH ? CELLS + 1 : 1];
    __shared__ int any_excess;
    // ... and a tile whose pairs beyond the fourth do not fit the tail table (a compressed plasma: 16 particles per cell
    // on average fill its 1024 entries) keeps no tail at all: every pair beyond a cell's fourth is an excess pair.  (What
    // the table could not take used to go to the deferred list and on to the global-atomics pass: 12 % of the particles
    // of the boosted wakefield deck, 41 ms per launch, profiles/round5/README.md.)
    __shared__ int tail_off;
    // ... and their data, for the first DKEEP entries of every bucket: phase C has the particle in registers when it
    // defers it; fetched again by index in phase D each one costs seven cache lines from HBM (the tile's lines have left
    // the L2 by then: FETCH_SIZE 1.57 x the particle data, phase D 12 % of the kernel for 3 % of the particles)
    // (half tiles: 8 per bucket = 7 KB, so that two workgroups of 79 KB fit a CU's 160 KB)
    constexpr int DKEEP = (FUSED ? 0 : TSZ != TS ? 8 : sizeof(ACC) == 8 ? 48 : 16) / (SNG ? 2 : 1);   // 16 x 48 x 56 B = 42 KB next to the 89 KB tile
    __shared__ double dkeep[7][FUSED ? 1 : NBKT * DKEEP];
    __shared__ double F[FUSED ? 6 * FUSED_GNPTS : 1];   // FUSED: Ex Ey Ez Bx By Bz of the tile + halo
    constexpr bool PT = CFG::PT != 0;
    static_assert(!PT || !FUSED, "persistent tiles: the deposition kernels");
    __shared__ long unit_s;
    const long ntiles = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == 0 && tid == 0 && sq.next) *sq.next = 0u;
    DPROF_INIT
    bool first_tile = true;        // PT: the tile has to be zeroed by phase A (later ones are left zeroed by the flush)
    unsigned xcd_done = 0;         // PT, thread 0: XCD ranges found exhausted
  int unit_u = 0, unit_k = 1;   // this workgroup takes the chunks ch = unit_u (mod unit_k) of its tile
  for (;;) {
    long unit;
    if constexpr (PT) {
        if (tid == 0) {
            const long nunits = ntiles * SUB, per = (nunits + 7) / 8;
            long u = -1;
            for (int t = 0; t < 8 && u < 0; ++t) {
                const int x = (int)((blockIdx.x + t) & 7);
                if (xcd_done & (1u << x)) continue;
                const long v = (long)atomicAdd(&tile_ctr[x], 1u);
                if (v < per && x * per + v < nunits) u = x * per + v;
                else xcd_done |= 1u << x;
            }
            unit_s = u;
        }
        __syncthreads();   // the claim; and the previous tile's flush (which reads and clears the tile) before this tile's phase A
        unit = unit_s;
        if (unit < 0) break;
    } else {
        // (a tile with far more particles than the others is shared by several workgroups: heavy_tiles.hpp)
        if (!heavy_unit_of(hu, blockIdx.x, ntiles * SUB, unit, unit_u, unit_k)) return;
    }
    // PT: the launch's uniform doubles are made opaque per tile.  Anything derived from them would otherwise be hoisted
    // out of the tile loop into VGPR pairs (there is no scalar fp64 unit) that stay live through every phase of every
    // tile: 168 VGPRs + 108 B of scratch against 151 without the loop.
    Geom g = g_;
    EsirkepovStep es = es_;
    double q = q_, relative_time = relative_time_;
    if constexpr (PT) {
        WXA_OPAQUE_UNIFORM_F64(g.xmin); WXA_OPAQUE_UNIFORM_F64(g.ymin); WXA_OPAQUE_UNIFORM_F64(g.zmin);
        WXA_OPAQUE_UNIFORM_F64(g.dxi); WXA_OPAQUE_UNIFORM_F64(g.dyi); WXA_OPAQUE_UNIFORM_F64(g.dzi);
        WXA_OPAQUE_UNIFORM_F64(es.t_half);
#pragma unroll
        for (int d = 0; d < 3; ++d) { WXA_OPAQUE_UNIFORM_F64(es.dtdx[d]); WXA_OPAQUE_UNIFORM_F64(es.invdtd[d]); }
        WXA_OPAQUE_UNIFORM_F64(q); WXA_OPAQUE_UNIFORM_F64(relative_time);
    }
    const long tile = CFG::TB ? blocked_tile(unit / SUB, tg.nt[0], tg.nt[1], tg.nt[2]) : unit / SUB;
    const int half = (int)(unit % SUB);
    const long ucell0 = tile * TILE_CELLS + half * CELLS;
    const int start = offsets[ucell0];
    const int end = offsets[ucell0 + CELLS];
    if (end <= start) {
        if constexpr (PT) { __syncthreads(); continue; }   // unit_s is read by everyone before thread 0 claims again
        else return;
    }
    constexpr int DCAP = DEFER / NBKT;
    auto defer = [&](const int ip, const int bank) {   // phase B: by index only (the particle has not been loaded)
        const int n = atomicAdd(&ndef[bank], 1);
        if (n < DCAP) deferred[bank * DCAP + n] = (unsigned)ip | 0x80000000u;
        else sq.push(ip);
    };
    // phase B's overflow cases (a cell with more than 2 RMAX particles, a full tail table) never pass through the chunk
    // loop.  Esirkepov: the deferred list (phase D's wide body).  Direct deposition: the straggler kernel -- phase D is
    // Esirkepov's body (until round 3 these particles went there whatever the algorithm: one cell in a million at 8 per
    // cell, seen first at 256^3).  The fused kernel, which pushes in the chunk loop: its own push + deposit list.
    auto defer_unloaded = [&](const int ip, const int bank) {
        if (unit_u != 0) return;   // a tile shared by several workgroups: what is deferred by index is unit 0's
        if constexpr (FUSED) fa.gq.push(ip);
        else if constexpr (CFG::ALGO == WXA_DEPOSIT_DIRECT) sq.push(ip);
        else defer(ip, bank);
    };
    auto defer_particle = [&](const int ip, const int bank, const ParticleState& pp) {
        const int n = atomicAdd(&ndef[bank], 1);
        if (n < DCAP) {
            const bool keep = n < DKEEP;
            deferred[bank * DCAP + n] = (unsigned)ip | (keep ? 0u : 0x80000000u);
            if (keep) {
                const int at = bank * DKEEP + n;
                dkeep[0][at] = pp.x; dkeep[1][at] = pp.y; dkeep[2][at] = pp.z; dkeep[3][at] = pp.w;
                dkeep[4][at] = pp.ux; dkeep[5][at] = pp.uy; dkeep[6][at] = pp.uz;
            }
        } else {
            sq.push(ip);
        }
    };
    constexpr int WAVES = NT / 64;
    // ---- A: cell counts, row masks; zero fill
    if (tid == 0) { nitems = 0; next_chunk = 0; any_excess = 0; tail_off = 0; }
    if (tid < NBKT) ndef[tid] = 0;
    int my_s = 0, my_n = 0, my_pairs = 0;
    unsigned long long my_mask[RT];   // wave-uniform: tail row r of this cell-wave
    int off_lo = 0, off_hi = 0;
    if constexpr (CFG::ZF != 0) {
        // ZF: the two offsets of the lane's cell are requested first and the tile is zeroed while they travel (as written
        // below, the wait for them stands in front of the zero fill: in-order issue)
        if (tid < CELLS) { off_lo = offsets[ucell0 + tid]; off_hi = offsets[ucell0 + tid + 1]; }
        if (!PT || first_tile)
            for (int a = tid; a < 3 * NPTS; a += NT) lds[a] = (ACC)0;
    }
    if (tid < CELLS) {
        my_s = CFG::ZF != 0 ? off_lo : offsets[ucell0 + tid];
        my_n = (CFG::ZF != 0 ? off_hi : offsets[ucell0 + tid + 1]) - my_s;
        cstart[tid] = my_s;
        if (tid == CELLS - 1) cstart[CELLS] = my_s + my_n;
        my_pairs = min((my_n + 1) >> 1, RMAX);
        if constexpr (!HF) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                my_mask[r] = __ballot(my_pairs > 4 + r);
                if (lane == 0) masks[r][wave] = my_mask[r];
            }
        }
    }
    if constexpr (CFG::ZF == 0) {
        if (!PT || first_tile)
            for (int a = tid; a < 3 * NPTS; a += NT) lds[a] = (ACC)0;
    }
    first_tile = false;
    if constexpr (FUSED) {
        // the six staggered components of the tile + halo, all of a lane's loads in flight before its first LDS write
        const int ti_ = (int)(tile % tg.nt[0]), tj_ = (int)((tile / tg.nt[0]) % tg.nt[1]);
        const int tk_ = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
        const int q0 = tg.cell_lo[0] + ti_ * TS + FUSED_GLO, q1 = tg.cell_lo[1] + tj_ * TS + FUSED_GLO;
        const int q2 = tg.cell_lo[2] + tk_ * TS + FUSED_GLO;
        constexpr int GN = FUSED_GN, PER = (FUSED_GNPTS + NT - 1) / NT;
        auto fetch = [&](const DevF& f, double (&r)[PER]) {
#pragma unroll
            for (int n = 0; n < PER; ++n) {
                const int a = tid + n * NT;
                const int i = q0 + a % GN, j = q1 + (a / GN) % GN, k = q2 + a / (GN * GN);
                const bool in = a < FUSED_GNPTS && i >= f.lo0 && i < f.lo0 + f.n0 && j >= f.lo1 && j < f.lo1 + f.n1 &&
                                k >= f.lo2 && k < f.lo2 + f.n2;
                r[n] = in ? f.p[f.off(i, j, k)] : 0.0;
            }
        };
        auto put = [&](int c, const double (&r)[PER]) {
#pragma unroll
            for (int n = 0; n < PER; ++n) {
                const int a = tid + n * NT;
                if (a < FUSED_GNPTS) F[c * FUSED_GNPTS + a] = r[n];
            }
        };
        double r0[PER], r1[PER], r2[PER], r3[PER], r4[PER], r5[PER];
        fetch(fa.Ex, r0); fetch(fa.Ey, r1); fetch(fa.Ez, r2); fetch(fa.Bx, r3); fetch(fa.By, r4); fetch(fa.Bz, r5);
        put(0, r0); put(1, r1); put(2, r2); put(3, r3); put(4, r4); put(5, r5);
    }
    __syncthreads();
    DPROF(0);
    // ---- B: scan of the 64 (row, cell-wave) counts, item table
    if constexpr (HF) {
        // lane p = (class b, member m) holds cell c = m BW + b: the MEM cells of a class are MEM consecutive lanes, the
        // ranks of their holes and of their pairs beyond the fourth are two segmented wave scans
        int c = 0, pairs = 4, holes = 0, over = 0, hpre = 0, opre = 0, htot = 0, otot = 0, cs = 0, cn = 0;
        const int b = tid / MEM, m = tid % MEM;
        if (tid < CELLS) {
            c = m * BW + b;
            cs = cstart[c];
            cn = cstart[c + 1] - cs;
            pairs = (cn + 1) >> 1;
            holes = max(0, 4 - pairs);
            over = min(max(0, pairs - 4), RT);
            int hi = holes, oi = over;
#pragma unroll
            for (int d = 1; d < MEM; d <<= 1) {
                const int vh = __shfl_up(hi, d, MEM), vo = __shfl_up(oi, d, MEM);
                if (m >= d) { hi += vh; oi += vo; }
            }
            htot = __shfl(hi, MEM - 1, MEM); otot = __shfl(oi, MEM - 1, MEM);
            hpre = hi - holes; opre = oi - over;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (t < over) {
                    const unsigned short ent = (unsigned short)(c | ((4 + t) << 9));
                    const int qq = opre + t;
                    if (qq < htot) {
                        fill[b * (4 * MEM) + qq] = ent;
                    } else {
                        const int at = atomicAdd(&nitems, 1);
                        if (at < TCAP) table[at] = ent;
                        else {   // any bucket is correct; the cell's place in the sort order is the bank of a particle that stayed
                            defer_unloaded(cs + 2 * (4 + t), c & (NBANK - 1));
                            if (2 * (4 + t) + 1 < cn) defer_unloaded(cs + 2 * (4 + t) + 1, c & (NBANK - 1));
                        }
                    }
                }
            }
            for (int k = 2 * RMAX; k < cn; ++k) defer_unloaded(cs + k, c & (NBANK - 1));   // beyond the table's rows (> 2 RMAX particles in a cell)
        }
        __syncthreads();
        if (tid < CELLS) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < holes) slot[pairs + t][c] = (hpre + t < otot) ? fill[b * (4 * MEM) + hpre + t] : NOFILL;
        }
    } else if (tid < CELLS) {
        // scan order of the 64 (row, cell-wave) counts: row-major, or -- TI -- cell-wave-major
        const int cnt = CFG::TI ? __popcll(masks[lane % RT][lane / RT]) : __popcll(masks[lane / CW][lane % CW]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int excl = incl - cnt;
        const int total = __shfl(incl, 63);
        const bool no_tail = XCH && total > TCAP;   // the same in every wave
        if (tid == 0) { nitems = no_tail ? 0 : min(total, TCAP); tail_off = no_tail; }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int base = __shfl(excl, CFG::TI ? wave * RT + r : r * CW + wave);   // first item of (tail row r, this cell-wave)
            if (my_pairs > 4 + r && !no_tail) {
                const int at = base + __popcll(my_mask[r] & lt);
                if (at < TCAP) table[at] = (unsigned short)(tid | ((4 + r) << 9));
                else {   // any bucket is correct; the cell's place in the sort order is the bank of a particle that stayed
                    defer_unloaded(my_s + 2 * (4 + r), tid & (NBANK - 1));
                    if (2 * (4 + r) + 1 < my_n) defer_unloaded(my_s + 2 * (4 + r) + 1, tid & (NBANK - 1));
                }
            }
        }
        // beyond the table's rows (> 2 RMAX particles in a cell): chunks of their own (XCH), or one by one to the lists
        if constexpr (XCH) {
            if (my_n > 2 * (no_tail ? 4 : RMAX)) any_excess = 1;
        } else {
            for (int k = 2 * RMAX; k < my_n; ++k) defer_unloaded(my_s + k, tid & (NBANK - 1));
        }
    }
    __syncthreads();
    int excess_pairs = 0;
    const int rmax_t = XCH && __builtin_amdgcn_readfirstlane(tail_off) ? 4 : RMAX;   // pairs of a cell that the direct chunks and the tail table cover
    if constexpr (XCH) {
        if (any_excess) {   // uniform
            if (tid < CELLS) xs[tid] = (max(0, my_n - 2 * rmax_t) + 1) >> 1;
            __syncthreads();
            if (tid == 0) {
                int acc = 0;
                for (int c = 0; c < CELLS; ++c) { const int v = xs[c]; xs[c] = acc; acc += v; }
                xs[CELLS] = acc;
            }
            __syncthreads();
            excess_pairs = __builtin_amdgcn_readfirstlane(xs[CELLS]);
        }
    }
    DPROF(1);
    const int ti = (int)(tile % tg.nt[0]);
    const int tj = (int)((tile / tg.nt[0]) % tg.nt[1]);
    const int tk = (int)(tile / ((long)tg.nt[0] * tg.nt[1]));
    const int o0 = tg.cell_lo[0] + ti * TS + TD::LO;
    const int o1 = tg.cell_lo[1] + tj * TS + TD::LO;
    const int o2 = tg.cell_lo[2] + tk * TS + half * TSZ + TD::LO;
    // ---- C: the chunks (NB blocks of 16 cells x 4 pairs, then the tail table)
    const int T = min(nitems, TCAP);
    constexpr bool COOP = CFG::COOP != 0 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && (O % 2) == 1 && sizeof(ACC) == 8;
    constexpr int TPC = COOP ? 32 : 64;   // tail items per chunk: with COOP the upper half of the wave only assists
#ifdef WXA_DEPOSIT_PROFILE
    const long long prof_l0 = clock64();
#endif
    const int CH0 = unit_u + unit_k * wave, CHS = unit_k * WAVES;   // the chunks of this unit, wave by wave
    const int nregular = NB + ((T + TPC - 1) / TPC);
    const int nchunks = CFG::DBG == 5 ? 0 : nregular + ((excess_pairs + 63) >> 6);   // DBG 5 (timing): the phases around the loop alone
    // DYN: a wave holds the chunk it works on and has already claimed the next one (the counter's round trip through the
    // LDS queue -- behind the other waves' atomics -- hides behind the chunk)
    auto claim = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(&next_chunk, 1);
        return v;
    };
    int claimed = 0;
    if constexpr (CFG::DYN != 0) claimed = claim();
    // the lane's work item of chunk ch: its two particles (an empty lane reads the tile's first particle)
    auto item_of = [&](const int q, int& ia, int& ib, bool& va, bool& vb) {
        int ch = q;
        if constexpr (CFG::TI != 0 && !HF) {   // the q-th chunk of the interleaved sequence
            const int ntail = nchunks - NB;
            const int tb = q * ntail / nchunks, ti = (q + 1) * ntail / nchunks;
            ch = ti > tb ? NB + tb : q - tb;
        }
        int c, r;
        if (ch < NB) {
            c = BW * (ch / (4 / RPC)) + (lane % BW); r = RPC * (ch % (4 / RPC)) + lane / BW; va = true;
            if constexpr (HF) {
                if (2 * r >= cstart[c + 1] - cstart[c]) {   // a hole: the pair of its class that fills it, if any
                    const unsigned ent = slot[r][c];
                    va = ent != NOFILL;
                    c = va ? (int)(ent & 511u) : c; r = va ? (int)(ent >> 9) : r;
                }
            }
        } else if (!XCH || ch < nregular) {
            const int I = (ch - NB) * TPC + (lane & (TPC - 1));
            va = I < T && lane < TPC;
            const unsigned ent = va ? table[I] : 0u;
            c = (int)(ent & 511u); r = (int)(ent >> 9);
        } else {   // pair I of the cells' excess: the cell whose running sum holds it, xs[c] <= I < xs[c + 1]
            // Lane (r, s) = (lane / 16, lane % 16) of excess chunk j takes pair 4 j + r of stream s, the streams being
            // sixteen equal ranges of the excess pairs: the 16 lanes that a step of a ds_add_f64 serves sit in sixteen
            // different parts of the tile (consecutive pairs -- one cell, one frame, 64 lanes on the same addresses --
            // made the excess chunks of a dense tile several times slower than its other chunks), and every stream
            // still reads four neighbouring pairs = 64 contiguous bytes per array.
            const int nexc = (excess_pairs + 63) >> 6;
            const int I = (lane & 15) * (4 * nexc) + 4 * (ch - nregular) + (lane >> 4);
            va = I < excess_pairs;
            int lo = 0, hi = CELLS;
            if constexpr (XCH) {
                const int J = va ? I : 0;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (xs[mid] <= J) lo = mid; else hi = mid;
                }
                r = rmax_t + (J - xs[lo]);
            } else {
                r = 0;
            }
            c = lo;
        }
        int s0, n0;
        if (CFG::GIDX != 0 && ch < NB) { s0 = offsets[ucell0 + c]; n0 = offsets[ucell0 + c + 1] - s0; }
        else { s0 = cstart[c]; n0 = cstart[c + 1] - s0; }
        va = va && 2 * r < n0;
        ia = va ? s0 + 2 * r : start;
        vb = va && 2 * r + 1 < n0;
        ib = vb ? ia + 1 : ia;
    };
    constexpr int PFD = (CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && !FUSED && !COOP && CFG::DYN == 0 && CFG::DBG == 0) ? CFG::PFD : 0;
    int nia = start, nib = start;
    bool nva = false, nvb = false, has_next = false;
    ParticleState na{}, nb{};
    auto request_next = [&]() {   // between two compiler fences: the loads stay where they are written
        if (has_next) {
            asm volatile("" ::: "memory");
            na = ParticleState{px[nia], py[nia], pz[nia], pw[nia], pux[nia], puy[nia], puz[nia]};
            nb = ParticleState{px[nib], py[nib], pz[nib], pw[nib], pux[nib], puy[nib], puz[nib]};
            asm volatile("" ::: "memory");
        }
    };
    if constexpr (PFD != 0) {
        has_next = CH0 < nchunks;
        if (has_next) item_of(CH0, nia, nib, nva, nvb);
        request_next();
    }
    for (int ch = CFG::DYN ? __builtin_amdgcn_readfirstlane(__shfl(claimed, 0)) : CH0; ch < nchunks;) {   // wave-uniform
        if constexpr (CFG::DYN != 0) claimed = claim();
        int ia, ib;
        bool va, vb;
        if constexpr (PFD != 0) {
            ia = nia; ib = nib; va = nva; vb = nvb;
            has_next = ch + CHS < nchunks;
            if (has_next) item_of(ch + CHS, nia, nib, nva, nvb);
        } else {
            item_of(ch, ia, ib, va, vb);
        }
        // all fourteen loads in flight together (an empty lane reads the tile's first particle)
#ifdef WXA_DEPOSIT_PROFILE
        const long long prof_c0 = clock64();
#endif
        ParticleState pa, pb;
        if constexpr (PFD != 0) {
            pa = na; pb = nb;
            if constexpr (PFD == 3) request_next();   // the next chunk's fourteen loads travel behind the whole body
        } else if (CFG::LD16 != 0 && end - start >= 2) {   // (a tile with one particle: below)
            typedef double D2U __attribute__((ext_vector_type(2), aligned(8)));
            const int b2 = ia + 1 < end ? ia : ia - 1;   // the pair's first index; a last particle without a partner comes second
            const D2U vx = *reinterpret_cast<const D2U*>(px + b2), vy = *reinterpret_cast<const D2U*>(py + b2);
            const D2U vz = *reinterpret_cast<const D2U*>(pz + b2), vw = *reinterpret_cast<const D2U*>(pw + b2);
            const D2U vux = *reinterpret_cast<const D2U*>(pux + b2), vuy = *reinterpret_cast<const D2U*>(puy + b2);
            const D2U vuz = *reinterpret_cast<const D2U*>(puz + b2);
            const bool first = ia == b2;
            pa = ParticleState{first ? vx.x : vx.y, first ? vy.x : vy.y, first ? vz.x : vz.y, first ? vw.x : vw.y,
                               first ? vux.x : vux.y, first ? vuy.x : vuy.y, first ? vuz.x : vuz.y};
            pb = vb ? ParticleState{vx.y, vy.y, vz.y, vw.y, vux.y, vuy.y, vuz.y} : pa;
        } else {
            pa = ParticleState{px[ia], py[ia], pz[ia], pw[ia], pux[ia], puy[ia], puz[ia]};
            pb = ParticleState{px[ib], py[ib], pz[ib], pw[ib], pux[ib], puy[ib], puz[ib]};
        }
#ifdef WXA_DEPOSIT_PROFILE   // wave 0 of every workgroup: cycles from the loads' issue to their arrival, and of the whole chunk
        long long prof_c1 = 0;
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prof_c1 = clock64();
        }
#endif
        if constexpr (FUSED) {
            // PushPX (PhysicalParticleContainer.cpp:2687-2785) of the lane's two particles from the staged field tile;
            // a particle whose gather stencil leaves the tile is pushed and deposited by the straggler kernels
            constexpr int GN = FUSED_GN, G1 = FUSED_GNPTS;
            const int q0 = o0 - TD::LO + FUSED_GLO, q1 = o1 - TD::LO + FUSED_GLO, q2 = o2 - TD::LO + FUSED_GLO;
            auto push_one = [&](ParticleState& pp, const int ip) -> bool {
                GatherShapes<O, 1> sh;
                gather_shapes<O, 1>(pp.x, pp.y, pp.z, fa.gg, sh);
                constexpr int NN = O + 1, NC = O;
                const int lo_i = min(sh.jn, sh.jc) - q0, hi_i = max(sh.jn + NN, sh.jc + NC) - 1 - q0;
                const int lo_j = min(sh.kn, sh.kc) - q1, hi_j = max(sh.kn + NN, sh.kc + NC) - 1 - q1;
                const int lo_k = min(sh.ln, sh.lc) - q2, hi_k = max(sh.ln + NN, sh.lc + NC) - 1 - q2;
                if (!(lo_i >= 0 && lo_j >= 0 && lo_k >= 0 && hi_i < GN && hi_j < GN && hi_k < GN)) return false;
                const int jn = sh.jn - q0, jc = sh.jc - q0, kn = sh.kn - q1, kc = sh.kc - q1, ln = sh.ln - q2, lc = sh.lc - q2;
#define WXA_FROWS(...) gather_rows_lds<__VA_ARGS__, GN, GN * GN, 2>
                double Exp = WXA_FROWS(NC, NN, NN)(F + 0 * G1 + jc + GN * (kn + GN * ln), sh.sxc, sh.syn, sh.szn);
                double Eyp = WXA_FROWS(NN, NC, NN)(F + 1 * G1 + jn + GN * (kc + GN * ln), sh.sxn, sh.syc, sh.szn);
                double Ezp = WXA_FROWS(NN, NN, NC)(F + 2 * G1 + jn + GN * (kn + GN * lc), sh.sxn, sh.syn, sh.szc);
                double Bzp = WXA_FROWS(NC, NC, NN)(F + 5 * G1 + jc + GN * (kc + GN * ln), sh.sxc, sh.syc, sh.szn);
                double Byp = WXA_FROWS(NC, NN, NC)(F + 4 * G1 + jc + GN * (kn + GN * lc), sh.sxc, sh.syn, sh.szc);
                double Bxp = WXA_FROWS(NN, NC, NC)(F + 3 * G1 + jn + GN * (kc + GN * lc), sh.sxn, sh.syc, sh.szc);
#undef WXA_FROWS
                WXA_OPAQUE_F64(Exp); WXA_OPAQUE_F64(Eyp); WXA_OPAQUE_F64(Ezp); WXA_OPAQUE_F64(Bxp); WXA_OPAQUE_F64(Byp); WXA_OPAQUE_F64(Bzp);
                push_momentum<CFG::PUSHER>(pp.ux, pp.uy, pp.uz, Exp, Eyp, Ezp, Bxp, Byp, Bzp, q, fa.m, fa.dt);
                update_position(pp.x, pp.y, pp.z, pp.ux, pp.uy, pp.uz, fa.dt);
                fa.p.ux[ip] = pp.ux; fa.p.uy[ip] = pp.uy; fa.p.uz[ip] = pp.uz;
                fa.p.x[ip] = pp.x; fa.p.y[ip] = pp.y; fa.p.z[ip] = pp.z;
                return true;
            };
            if (va && !push_one(pa, ia)) { fa.gq.push(ia); va = false; }
            if (vb && !push_one(pb, ib)) { fa.gq.push(ib); vb = false; }
        }
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
            if constexpr (CFG::DYN != 0) ch = __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
            else ch += CHS;
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
            if constexpr (CFG::DYN != 0) ch = __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
            else ch += CHS;
            continue;
        }
        if constexpr (CFG::DBG == 4) {   // timing: the loop's items and loads alone
            if (pa.x + pa.y + pa.z + pa.w + pa.ux + pa.uy + pa.uz + pb.x + pb.y + pb.z + pb.w + pb.ux + pb.uy + pb.uz == 1.2345e-300) lds[0] = (ACC)pa.x;
            ch += CHS;
            continue;
        }
        EsirkepovCoords c1 = esirkepov_coords(pa, g, es), c2 = esirkepov_coords(pb, g, es);
        if constexpr (CFG::WL != 0 && CFG::ALGO == WXA_DEPOSIT_ESIRKEPOV && !FUSED) {
            // The chunks of the cells' excess pairs (a density spike: 10^3 .. 10^5 particles in a cell) are 64 lanes on one or
            // two cells: the lanes that share a frame -- hu.wave_sum_min or more of them -- sum every value over the wave
            // (wave_sum_f64, VALU only) and one lane adds it, instead of 64 lanes adding to one LDS address one after
            // the other.  Wave-uniform control flow throughout: every lane computes, with weight 0 where it has no part.
            const bool crowded = XCH && ch >= nregular;   // wave-uniform
            const WideFrame<O> fa = esirkepov_wide_frame<O>(c1, g), fb = esirkepov_wide_frame<O>(c2, g);
            auto fits = [&](const WideFrame<O>& f) {
                const int wi = f.b[0] - o0, wj = f.b[1] - o1, wk = f.b[2] - o2;
                return wi >= 0 && wj >= 0 && wk >= 0 && wi + O + 2 <= N && wj + O + 2 <= N && wk + O + 2 <= NZ;
            };
            bool on_a = va && fits(fa), on_b = vb && fits(fb);
            if (va && !on_a) sq.push(ia);   // the frame leaves the tile: the global-atomics pass
            if (vb && !on_b) sq.push(ib);
            const double wqa = q * pa.w, wqb = q * pb.w;
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
            const bool merged = on_a && on_b && fa.b[0] == fb.b[0] && fa.b[1] == fb.b[1] && fa.b[2] == fb.b[2];
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
            if (on_a) {
                const double wq2 = merged ? wqb : 0.0;
                auto component = [&](auto comp) {
                    fence();
                    const WideFrame<O> f1 = unpack_wide_frame<O>(qa, g), f2 = unpack_wide_frame<O>(qb, g);
                    LdsSink<M, TSZ, ACC> sink(lds, f1.b[0] - o0, f1.b[1] - o1, f1.b[2] - o2);
                    esirkepov_pair_wide<O, decltype(comp)::value>(c1, f1, wqa, c2, f2, wq2, es, sink);
                };
                component(std::integral_constant<int, 0>{});
                component(std::integral_constant<int, 1>{});
                component(std::integral_constant<int, 2>{});
            }
            if (on_b && !merged) {
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
            if constexpr (CFG::DYN != 0) ch = __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
            else ch += CHS;
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
        if constexpr ((O & 1) == 0 && CFG::SNG != 0 && !FUSED && !COOP && CFG::DBG == 0) {
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
            if constexpr (CFG::DYN != 0) ch = __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
            else ch += CHS;
            continue;
        }
        int key = -1;
        bool fast_a = false, fast_b = false;   // which of the lane's particles its fast item holds
        if (sa == 0) {
            key = ka; fast_a = true;
            if (sb == 0 && kb == ka) {
                wq2 = wqb; fast_b = true;                // merged with its neighbour
            } else {
                if (sb == 0) {   // another frame than its lane partner: deferred, alone
                    if constexpr (SNG) defer_particle(ib, NBANK + (((kb & 15) + 8 * (kb >> 8)) & (NBANK - 1)), pb);   // its fast frame's bank
                    else defer_particle(ib, wide_bank(c2), pb);   // the wide body takes it
                }
                c2 = c1;                                 // empty partner
            }
        } else if (sb == 0) {
            key = kb; c1 = c2; wq1 = wqb; fast_b = true;   // the second particle alone
        }
        if constexpr (COOP) {
            // lanes l and l + 32 work on one frame: a lane without a fast item assists with zero weights; two different
            // frames (a stale sort: one of the pairs has left the cell) -- the upper lane hands its particles to phase D
            const int pkey = partner32(key);
            if (pkey >= 0 && pkey != key) {
                if (key < 0 || lane >= 32) {
                    if (key >= 0) {
                        if (fast_a) defer_particle(ia, wide_bank(esirkepov_coords(pa, g, es)), pa);
                        if (fast_b) defer_particle(ib, wide_bank(esirkepov_coords(pb, g, es)), pb);
                    }
                    key = pkey; wq1 = 0.0; wq2 = 0.0; c2 = c1;
                }
            }
            if (key >= 0) {
                LdsSink<M, TSZ, ACC> sink(lds, key & 15, (key >> 4) & 15, key >> 8);
                constexpr int H = (O + 1) / 2;
                LdsSink<M, TSZ, ACC> sink_jxy = lane >= 32 ? sink.shifted(0, H) : sink;
                LdsSink<M, TSZ, ACC> sink_jz = lane >= 32 ? sink.shifted(H, 0) : sink;
                esirkepov_pair_phased_coop<O>(c1, c2, wq1, wq2, es, sink_jxy, sink_jz);
            }
        } else if (key >= 0) {
            LdsSink<M, TSZ, ACC> sink(lds, key & 15, (key >> 4) & 15, key >> 8);
            if constexpr (CFG::DBG == 1) {
                NullSink ns;
                esirkepov_pair_phased<O, CFG::PHASED == 2>(c1, c2, wq1, wq2, es, ns);
                if (ns.acc == 1.2345e-300) lds[0] = (ACC)ns.acc;
            } else if constexpr (CFG::DBG == 3) {
                if (c1.x_new + c2.x_new + wq1 + wq2 == 1.2345e-300) lds[0] = (ACC)wq1;   // neither: the loop's skeleton (loads, coordinates, frames, deferrals)
            } else if constexpr (CFG::DBG == 2) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                    for (int b = 1; b <= O + 1; ++b)
#pragma unroll
                        for (int a = 1; a <= O + 1; ++a)
#pragma unroll
                            for (int l = 1; l <= O; ++l) sink.add(cc, l, a, b, wq1 + wq2);
            } else if constexpr (PFD == 2) {
                esirkepov_pair_phased<O, CFG::PHASED == 2>(c1, c2, wq1, wq2, es, sink, request_next, NoHook{});
            } else if constexpr (PFD == 1) {
                esirkepov_pair_phased<O, CFG::PHASED == 2>(c1, c2, wq1, wq2, es, sink, NoHook{}, request_next);
            } else {
                esirkepov_pair_phased<O, CFG::PHASED == 2>(c1, c2, wq1, wq2, es, sink);
            }
        } else if constexpr (PFD == 1 || PFD == 2) {
            request_next();   // a lane without a fast item
        }
#ifdef WXA_DEPOSIT_PROFILE
        if (wave == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the chunk's atomics have left the queue
            const long long prof_c2 = clock64();
            if (tid == 0) { DCOUNT(6, prof_c1 - prof_c0); DCOUNT(7, prof_c2 - prof_c1); DCOUNT(8, 1); }
        }
#endif
        if constexpr (CFG::DYN != 0) ch = __builtin_amdgcn_readfirstlane(__shfl(claimed, 0));
        else ch += CHS;
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
    // PT: the three array descriptors (36 SGPRs) are read from the kernel-argument segment here, per tile: as loop
    // invariants they stayed in SGPRs through the whole tile loop, the kernel ran out of them (106) and spilled into VGPRs
    const JTriple* jt = &J3;
    if constexpr (PT) jt = WXA_LATE_KERNARG(JTriple, J3);
    if constexpr (CFG::FLUSH == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const DevF J = c == 0 ? jt->x : c == 1 ? jt->y : jt->z;
            for (int a = tid; a < NPTS; a += NT) {
                const double v = (double)lds[c * NPTS + a];
                if (v != 0.0) {
                    if constexpr (PT) lds[c * NPTS + a] = (ACC)0;
                    const int i = o0 + (a % PS) % TD::NS, j = o1 + (a % PS) / TD::NS, k = o2 + a / PS;
                    if (i >= J.lo0 && i < J.lo0 + J.n0 && j >= J.lo1 && j < J.lo1 + J.n1 && k >= J.lo2 &&
                        k < J.lo2 + J.n2)
                        atomic_add_f64(J.p + J.off(i, j, k), v);
                }
            }
        }
    } else {
        // FLUSH = 1: a lane owns one (i, j) column of one component and walks it along k -- all NZ LDS reads of the
        // column in flight at once, one address computation per column instead of one 64-bit index decomposition per
        // point.  (As written above, per point: ds_read, s_waitcnt lgkmcnt(0), compare, ~20 integer instructions of
        // address arithmetic, atomic -- fifteen times in a row per lane, 7 % of the kernel.)
        constexpr int NSC = TD::NS, COLS = N * NSC;   // columns of a component, the padding points of a row included
        for (int col = tid; col < 3 * COLS; col += NT) {
            const int c = col / COLS, ij = col - c * COLS;
            const int li = ij % NSC, lj = ij / NSC;
            ACC* src = lds + c * NPTS + ij;
            double v[NZ];
#pragma unroll
            for (int k = 0; k < NZ; ++k) v[k] = (double)src[k * PS];
            if constexpr (PT) {
#pragma unroll
                for (int k = 0; k < NZ; ++k) src[k * PS] = (ACC)0;
            }
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
    if constexpr (!PT) break;
  }
    DPROF_FINISH
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
    const long nunits = (long)tg.nt[0] * tg.nt[1] * tg.nt[2] * (TS / CFG::TSZ);
    const Geom g = make_geom(*geom);
    const int* offsets = (const int*)ws->offsets.p;
    // persistent tiles: one workgroup per CU (WXA_NUM_CU; more would only queue behind the LDS), tiles through tile_ctr
    wxa_status rc;
    // tiles with far more particles than the others are shared by several workgroups (heavy_tiles.hpp)
    HeavyUnits hu;
    long extra_groups = 0;
    if constexpr (CFG::PT == 0 && CFG::DYN == 0 && CFG::FUSED == 0 && CFG::TSZ == TS) {
        if ((rc = plan_heavy_tiles(ws, offsets, nunits, (long)p->np, hu, extra_groups, st)) != WXA_OK) return rc;
    }
    const dim3 grid((unsigned)(CFG::PT ? std::min<long>(xcd_grid_size(nunits), WXA_NUM_CU) : xcd_grid_size(nunits) + extra_groups)), block(CFG::NT);
    if ((rc = ws->stragglers.reserve(sizeof(int) * (size_t)p->np + 64)) != WXA_OK) return rc;
    StragglerQueue sq{(int*)ws->stragglers.p, nullptr, nullptr};
    unsigned *cnt_now = nullptr, *cnt_next = nullptr;
    if ((rc = flip_counter(ws, 0, ws->deposit_flips, st, cnt_now, cnt_next)) != WXA_OK) return rc;
    sq.count = cnt_now; sq.next = cnt_next;   // words 0, 1 of ws->counters
    unsigned* tile_ctr = (unsigned*)ws->counters.p + 96;   // words of ws->counters: see particles.hip (0, 1 deposit, 16, 17 gather, 32 classify, 48 walls, 56 injection, 64..90 destinations); 96..103: tile claims per XCD
    if (CFG::PT) WXA_HIP_CHECK(hipMemsetAsync(tile_ctr, 0, 8 * sizeof(unsigned), st));
    const DevF jx = make_devf(J[0]), jy = make_devf(J[1]), jz = make_devf(J[2]);
    const EsirkepovStep es = make_esirkepov_step(g, dt, relative_time);
    hipLaunchKernelGGL((deposit_tile_rows_kernel<O, MARGIN, CFG>), grid, block, 0, st, JTriple{jx, jy, jz}, p->x, p->y, p->z, p->w,
                       p->ux, p->uy, p->uz, offsets, g, tg, q, es, relative_time, sq, FusedArgs{}, tile_ctr, hu);
    hipLaunchKernelGGL((deposit_stragglers_kernel<O, CFG::ALGO>), dim3(512), dim3(256), 0, st, p->x, p->y,
                       p->z, p->w, p->ux, p->uy, p->uz, sq.idx, sq.count, jx, jy, jz, g, q, es, relative_time);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

// Production configurations (whole tiles, 768 lanes = 3 waves per SIMD, blocks of 32 cells x 2 pairs): Esirkepov, Esirkepov
// on fp32 tiles (opt-in per workspace), direct deposition.  Measured and removed again, ms per launch at 256^3 x 8 ppc
// against 6.3 (profiles/round2/README.md, profiles/round3/README.md): the staged / bucketed kernel of round 1 9.4; half
// tiles with 2 x 6 waves 9.1 and 2 x 8 waves at 128 VGPRs 7.0; whole tiles with 16 waves at 128 VGPRs 7.7; an L2 prefetch
// of the next chunk +0.35; blocks of 16 cells x 4 pairs 6.6 (the same addresses in consecutive 16-lane steps of a
// ds_add_f64 cost 11 cycles per wave instruction instead of 8, scripts/microbench/lds_atomic_bench.hip); lanes l and l + 32
// sharing their deposits through v_permlane32_swap (half the LDS atomics, + 17 % VALU) 6.6 -- kept as dev variant 22.
// Round 4: the write-back by columns (FLUSH = 1) 6.06-6.13 ms against 6.11-6.20 in four interleaved repeats
// (profiles/round4/r4j_deposit_flush_by_columns.txt); everything else measured in round 4 (hole filling, persistent tiles,
// chunk offsets from global memory, half tiles, the work item read ahead) did not beat this configuration and stays a dev variant.
// ... then, same round (r4m, r4n: four interleaved repeats each): the zero fill behind the loads of the cell offsets (ZF)
// 5.98 against 6.04, the lone partners of phase D on their own fast frame (SNG) 5.94, both 5.88-5.96 against 6.01-6.05.
// ... and the last sessions of the round (u - y, profiles/round4/README.md): timing builds took the kernel apart (complete 5.9,
// without the pair body 3.8, items and loads alone 2.8, the phases around the loop 0.74 ms) and five more candidates did not
// beat it: the next chunk's particles requested mid-body (PFD), the tail's chunks interleaved by cell (TI), tiles in blocks
// (TB), 16-byte particle loads (LD16), 16 waves at 128 VGPRs.  Direct deposition (SNG = 1 there: the lane's two particles on
// one frame per component, the second one deferred when it sits on another frame): 14.8 -> 7.5 ms.
using RowsEsirkepov = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;
using RowsEsirkepovF32 = RowsCfg<768, 8, 3, 1, 0, float, 0, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // fp32 tile accumulation (ds_add_f32), opt-in per workspace
using RowsDirect = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_DIRECT, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;
using RowsEsirkepovStreaming = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 1>;   // WL
using RowsDirectSeq = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_DIRECT, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1>;   // 85 (dev builds): the lane's two particles one after the other, as until round 4 (32-cell chunks: 16.1 -> 14.9 ms in round 3)
#ifdef WXA_DEV_VARIANTS   // A/B timing builds only (scripts/variants.py): WXA_DEPOSIT_VARIANT=<n>, order-3 Esirkepov
using RowsB16 = RowsCfg<768, 8, 3, 1, 0, double, 16>;
using RowsB16Coop = RowsCfg<768, 8, 3, 1, 0, double, 16, WXA_DEPOSIT_ESIRKEPOV, 1>;
using RowsB32Coop = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 1>;
using RowsW10 = RowsCfg<640, 8, 3, 1, 0, double, 32>;   // 30: 10 waves -- 38 chunks of a tile in 4 rounds of 10 instead of 12
using RowsW11 = RowsCfg<704, 8, 3, 1, 0, double, 32>;   // 31: 11 waves
using RowsDyn = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 1>;   // 40: chunks through an LDS counter
using RowsHFPT = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 1, 1>;   // 62: hole filling + persistent tiles
using RowsPT = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 1>;   // 63: persistent tiles alone
using RowsHF = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 1, 0>;   // 64: hole filling alone
using RowsGIdx = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 1>;   // 65: chunk indices from global memory
using RowsGIdxDyn = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 1, 0, WXA_PUSHER_BORIS, 0, 0, 1>;   // 66: ... + dynamic chunks
using RowsHalf6 = RowsCfg<384, 4, 3, 1, 0, double, 32>;   // 70: half tiles (8 x 8 x 4 cells, 79 KB of LDS), two workgroups of 6 waves per CU
using RowsHalf8 = RowsCfg<512, 4, 4, 1, 0, double, 32>;   // 71: ... of 8 waves at 128 VGPRs
using RowsFlushPoints = RowsCfg<768, 8, 3, 1, 0, double, 32>;   // 80: production until round 3 (the write-back point by point)
using RowsZeroFirst = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1>;   // 81
using RowsSingles = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 0, 1>;   // 82: lone partners on their fast frame in phase D
using RowsRound4J = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1>;   // 83: production of session j (write-back by columns only)
using RowsW16 = RowsCfg<1024, 8, 4, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 90: production with 16 waves at 128 VGPRs
using RowsPfd1 = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 1>;   // 91: next chunk's particles requested before the last component
using RowsPfd2 = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 2>;   // 92: ... before the second
using RowsTailInterleaved = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 0, 1>;   // 93: the tail's chunks among the direct ones
using RowsTileBlocks = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 0, 0, 1>;   // 94: tiles in blocks of 4 x 4 x 2
using RowsLd16 = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1>;   // 95: 16-byte particle loads
using RowsLd16LoadsOnly = RowsCfg<768, 8, 3, 1, 4, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1>;   // 116: 114 with them
using RowsHFDyn = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 1, 0, WXA_PUSHER_BORIS, 1>;   // 61: ... + dynamic chunks
using RowsNoLds = RowsCfg<768, 8, 3, 1, 1, double, 32>;   // 101: the arithmetic without the LDS atomics (wrong J)
using RowsNoAlu = RowsCfg<768, 8, 3, 1, 2, double, 32>;   // 102: the LDS atomics without the arithmetic (wrong J)
using RowsNoLdsNew = RowsCfg<768, 8, 3, 1, 1, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 111: 101 on the production configuration
using RowsNoAluNew = RowsCfg<768, 8, 3, 1, 2, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 112: 102 on it
using RowsLoadsOnly = RowsCfg<768, 8, 3, 1, 4, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 114
using RowsPhasesOnly = RowsCfg<768, 8, 3, 1, 5, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 115
using RowsSkeleton = RowsCfg<768, 8, 3, 1, 3, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;   // 113: neither (wrong J)
// round 6: two waves per SIMD with the whole register file (256 VGPRs): room for the next chunk's particles in registers
using RowsW8 = RowsCfg<512, 8, 2, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1>;            // 120
using RowsW8Pfd1 = RowsCfg<512, 8, 2, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 1>;     // 121
using RowsW8Pfd2 = RowsCfg<512, 8, 2, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 2>;     // 122
using RowsW8Pfd3 = RowsCfg<512, 8, 2, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 0, WXA_PUSHER_BORIS, 0, 0, 0, 1, 1, 1, 3>;     // 123: requested at the top of the chunk
#endif

#ifdef WXA_DEV_VARIANTS   // measured and not adopted (wxa_debug_push_and_deposit, particles.hip): 14.8 ms against 4.7 + 6.8
// PushPX + DepositCurrent of the sorted part of a tile in one kernel (CFG::FUSED), then the two straggler lists: particles
// whose gather stencil left the staged field tile (pushed, then deposited) and pushed particles whose deposit left the J tile.
using RowsFusedBoris = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 1, WXA_PUSHER_BORIS>;
using RowsFusedVay = RowsCfg<768, 8, 3, 1, 0, double, 32, WXA_DEPOSIT_ESIRKEPOV, 0, 0, 1, WXA_PUSHER_VAY>;

bool push_deposit_tile_available(const wxa_workspace* ws, const wxa_particle_view* p, int order, int galerkin, int pusher,
                                 int algo) {
    if (!deposit_tile_available(ws, p) || order != 3 || !galerkin || algo != WXA_DEPOSIT_ESIRKEPOV) return false;
    if (pusher != WXA_PUSHER_BORIS && pusher != WXA_PUSHER_VAY) return false;
    if (ws->deposit_accumulator != WXA_ACC_FP64 || ws->lens_n > 0) return false;
    for (int c = 0; c < 6; ++c)
        if (ws->ext_eb[c] != 0.0) return false;
    return true;
}

template <class CFG>
static wxa_status launch_fused(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                               const wxa_field_view J[3], const wxa_grid_geom* geom_eb, const wxa_grid_geom* geom_j, double q,
                               double m, double dt, double relative_time, wxa_workspace* ws, hipStream_t st) {
    TileGeom tg;
    for (int d = 0; d < 3; ++d) {
        tg.nt[d] = (ws->sort_nc[d] + TS - 1) / TS;
        tg.cell_lo[d] = ws->sort_cell_lo[d];
    }
    const long nunits = (long)tg.nt[0] * tg.nt[1] * tg.nt[2];
    const Geom g = make_geom(*geom_j);
    const int* offsets = (const int*)ws->offsets.p;
    const dim3 grid((unsigned)xcd_grid_size(nunits)), block(CFG::NT);
    wxa_status rc;
    if ((rc = ws->stragglers.reserve(2 * (sizeof(int) * (size_t)p->np + 64))) != WXA_OK) return rc;
    if ((rc = ws->counters.reserve(512)) != WXA_OK) return rc;
    StragglerQueue sq{(int*)ws->stragglers.p, (unsigned*)ws->counters.p};
    FusedArgs fa;
    fa.p = make_pv(*p);
    fa.Ex = make_devf(E[0]); fa.Ey = make_devf(E[1]); fa.Ez = make_devf(E[2]);
    fa.Bx = make_devf(B[0]); fa.By = make_devf(B[1]); fa.Bz = make_devf(B[2]);
    fa.gg = make_geom(*geom_eb);
    fa.m = m; fa.dt = dt;
    fa.gq = StragglerQueue{(int*)ws->stragglers.p + p->np + 16, (unsigned*)ws->counters.p + 16};
    WXA_HIP_CHECK(hipMemsetAsync(sq.count, 0, sizeof(unsigned), st));
    WXA_HIP_CHECK(hipMemsetAsync(fa.gq.count, 0, sizeof(unsigned), st));
    const DevF jx = make_devf(J[0]), jy = make_devf(J[1]), jz = make_devf(J[2]);
    const EsirkepovStep es = make_esirkepov_step(g, dt, relative_time);
    hipLaunchKernelGGL((deposit_tile_rows_kernel<3, MARGIN, CFG>), grid, block, 0, st, JTriple{jx, jy, jz}, p->x, p->y, p->z, p->w,
                       p->ux, p->uy, p->uz, offsets, g, tg, q, es, relative_time, sq, fa, (unsigned*)nullptr, HeavyUnits{});
    // the particles the tile kernel could not push: global-memory gather + push, then their deposit
    if ((rc = gather_push_listed(p, fa.gq.idx, fa.gq.count, E, B, geom_eb, q, m, dt, CFG::PUSHER, st)) != WXA_OK) return rc;
    hipLaunchKernelGGL((deposit_stragglers_kernel<3, WXA_DEPOSIT_ESIRKEPOV>), dim3(512), dim3(256), 0, st, p->x, p->y,
                       p->z, p->w, p->ux, p->uy, p->uz, fa.gq.idx, fa.gq.count, jx, jy, jz, g, q, es, relative_time);
    hipLaunchKernelGGL((deposit_stragglers_kernel<3, WXA_DEPOSIT_ESIRKEPOV>), dim3(512), dim3(256), 0, st, p->x, p->y,
                       p->z, p->w, p->ux, p->uy, p->uz, sq.idx, sq.count, jx, jy, jz, g, q, es, relative_time);
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}

wxa_status push_deposit_tiled(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                              const wxa_field_view J[3], const wxa_grid_geom* geom_eb, const wxa_grid_geom* geom_j, double q,
                              double m, double dt, double relative_time, int pusher, wxa_workspace* ws, hipStream_t st) {
    if (pusher == WXA_PUSHER_VAY) return launch_fused<RowsFusedVay>(p, E, B, J, geom_eb, geom_j, q, m, dt, relative_time, ws, st);
    return launch_fused<RowsFusedBoris>(p, E, B, J, geom_eb, geom_j, q, m, dt, relative_time, ws, st);
}
#endif

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
#ifdef WXA_DEV_VARIANTS
        if (const char* e = getenv("WXA_DEPOSIT_VARIANT")) {
            switch (atoi(e)) {
                case 14: return launch_rows<3, RowsB16>(p, J, geom, q, dt, relative_time, ws, st);
                case 20: return launch_rows<3, RowsB16Coop>(p, J, geom, q, dt, relative_time, ws, st);
                case 22: return launch_rows<3, RowsB32Coop>(p, J, geom, q, dt, relative_time, ws, st);
                case 30: return launch_rows<3, RowsW10>(p, J, geom, q, dt, relative_time, ws, st);
                case 31: return launch_rows<3, RowsW11>(p, J, geom, q, dt, relative_time, ws, st);
                case 40: return launch_rows<3, RowsDyn>(p, J, geom, q, dt, relative_time, ws, st);
                case 62: return launch_rows<3, RowsHFPT>(p, J, geom, q, dt, relative_time, ws, st);
                case 63: return launch_rows<3, RowsPT>(p, J, geom, q, dt, relative_time, ws, st);
                case 64: return launch_rows<3, RowsHF>(p, J, geom, q, dt, relative_time, ws, st);
                case 65: return launch_rows<3, RowsGIdx>(p, J, geom, q, dt, relative_time, ws, st);
                case 66: return launch_rows<3, RowsGIdxDyn>(p, J, geom, q, dt, relative_time, ws, st);
                case 70: return launch_rows<3, RowsHalf6>(p, J, geom, q, dt, relative_time, ws, st);
                case 71: return launch_rows<3, RowsHalf8>(p, J, geom, q, dt, relative_time, ws, st);
                case 80: return launch_rows<3, RowsFlushPoints>(p, J, geom, q, dt, relative_time, ws, st);
                case 81: return launch_rows<3, RowsZeroFirst>(p, J, geom, q, dt, relative_time, ws, st);
                case 82: return launch_rows<3, RowsSingles>(p, J, geom, q, dt, relative_time, ws, st);
                case 83: return launch_rows<3, RowsRound4J>(p, J, geom, q, dt, relative_time, ws, st);
                case 90: return launch_rows<3, RowsW16>(p, J, geom, q, dt, relative_time, ws, st);
                case 91: return launch_rows<3, RowsPfd1>(p, J, geom, q, dt, relative_time, ws, st);
                case 92: return launch_rows<3, RowsPfd2>(p, J, geom, q, dt, relative_time, ws, st);
                case 93: return launch_rows<3, RowsTailInterleaved>(p, J, geom, q, dt, relative_time, ws, st);
                case 94: return launch_rows<3, RowsTileBlocks>(p, J, geom, q, dt, relative_time, ws, st);
                case 95: return launch_rows<3, RowsLd16>(p, J, geom, q, dt, relative_time, ws, st);
                case 116: return launch_rows<3, RowsLd16LoadsOnly>(p, J, geom, q, dt, relative_time, ws, st);
                case 61: return launch_rows<3, RowsHFDyn>(p, J, geom, q, dt, relative_time, ws, st);
                case 101: return launch_rows<3, RowsNoLds>(p, J, geom, q, dt, relative_time, ws, st);
                case 102: return launch_rows<3, RowsNoAlu>(p, J, geom, q, dt, relative_time, ws, st);
                case 111: return launch_rows<3, RowsNoLdsNew>(p, J, geom, q, dt, relative_time, ws, st);
                case 112: return launch_rows<3, RowsNoAluNew>(p, J, geom, q, dt, relative_time, ws, st);
                case 114: return launch_rows<3, RowsLoadsOnly>(p, J, geom, q, dt, relative_time, ws, st);
                case 115: return launch_rows<3, RowsPhasesOnly>(p, J, geom, q, dt, relative_time, ws, st);
                case 113: return launch_rows<3, RowsSkeleton>(p, J, geom, q, dt, relative_time, ws, st);
                case 120: return launch_rows<3, RowsW8>(p, J, geom, q, dt, relative_time, ws, st);
                case 121: return launch_rows<3, RowsW8Pfd1>(p, J, geom, q, dt, relative_time, ws, st);
                case 122: return launch_rows<3, RowsW8Pfd2>(p, J, geom, q, dt, relative_time, ws, st);
                case 123: return launch_rows<3, RowsW8Pfd3>(p, J, geom, q, dt, relative_time, ws, st);
                default: break;
            }
        }
#endif
        return launch_rows<3, RowsEsirkepov>(p, J, geom, q, dt, relative_time, ws, st);
    }
#ifdef WXA_DEV_VARIANTS
    if (const char* e = getenv("WXA_DEPOSIT_VARIANT"); e && atoi(e) == 85 && order == 3)
        return launch_rows<3, RowsDirectSeq>(p, J, geom, q, dt, relative_time, ws, st);
#endif
    if (order == 1) return launch_rows<1, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
    if (order == 2) return launch_rows<2, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
    return launch_rows<3, RowsDirect>(p, J, geom, q, dt, relative_time, ws, st);
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
