// The cell sort folded into PushPX (wxa_push_sort_begin / _end, include/warpx_amd.h).
//
// amrex's SortParticlesByBin (MultiParticleContainer.cpp:615-621) only permutes a tile, and the push has every particle
// in registers anyway.  So instead of a pass that reads the positions again to key them (sort_count_kernel) and a pass
// that moves all eight arrays (sort_scatter_window_kernel: 152 B per particle, 20 GB at 256^3 x 8 per cell):
//   * COUNT   -- the push of the step BEFORE a sort keys the particle's new position (tile-major cell key, the periodic
//                directions wrapped the way Redistribute will wrap the position) and takes its rank among equal keys;
//                the histogram is scanned when the push is done;
//   * SCATTER -- the push of the sort step writes position and momentum to the destination tile at offsets[key] + rank
//                instead of in place, and carries weight and id over.
// The order that comes out is the cell order of the positions BEFORE the sort step's push: one push older than a sort
// behind the push, which the LDS-tile kernels tolerate by construction (stale sorts: stragglers, deferred lists).
// Particles retired since the count stay where they were counted (dead weight until the next cycle); particles appended
// since the count keep their order behind the cell-sorted ones.
#ifndef WXA_PUSH_SORT_HPP_
#define WXA_PUSH_SORT_HPP_

#include "gather_body.hpp"
#include "workspace.hpp"

namespace wxa {

// ---- counting sort by cell: the key ---------------------------------------------------
struct SortGeom {
    double plo[3];
    double dinv[3];
    int nc[3];
    int retired_bin;   // key of retired particles (= number of cell bins): they end up behind the live ones
    int wrap[3] = {0, 0, 0};   // a cell index one period outside is brought back (the position is wrapped later); else clamped
};

// Tile-major cell key: tiles of WXA_TILE^3 cells, so that a tile's particles are contiguous
// (LDS-tile kernels) and still grouped by cell.  Inside a tile the order is i fastest, then the
// parity of k, then j, then k/2: any 16 consecutive cells (8 i x 2 k-parities) start on 16
// different LDS banks of the deposition tile (deposit_tile.hip, plane stride = 8 mod 16), which
// keeps its bank buckets evenly filled.
__device__ __forceinline__ int cell_of(const SortGeom& s, double x, double y, double z) {
    int i = (int)floor((x - s.plo[0]) * s.dinv[0]);
    int j = (int)floor((y - s.plo[1]) * s.dinv[1]);
    int k = (int)floor((z - s.plo[2]) * s.dinv[2]);
    if (s.wrap[0]) i = i < 0 ? i + s.nc[0] : (i >= s.nc[0] ? i - s.nc[0] : i);
    if (s.wrap[1]) j = j < 0 ? j + s.nc[1] : (j >= s.nc[1] ? j - s.nc[1] : j);
    if (s.wrap[2]) k = k < 0 ? k + s.nc[2] : (k >= s.nc[2] ? k - s.nc[2] : k);
    i = min(max(i, 0), s.nc[0] - 1);
    j = min(max(j, 0), s.nc[1] - 1);
    k = min(max(k, 0), s.nc[2] - 1);
    constexpr int T = WXA_TILE;
    const int nti = (s.nc[0] + T - 1) / T, ntj = (s.nc[1] + T - 1) / T;
    const int tile = (i / T) + nti * ((j / T) + ntj * (k / T));
    const int kt = k % T;
    return tile * (T * T * T) + (i % T) + T * ((kt & 1) + 2 * ((j % T) + T * (kt >> 1)));
}

constexpr int PUSH_SORT_COUNT = WXA_PUSH_SORT_COUNT, PUSH_SORT_SCATTER = WXA_PUSH_SORT_SCATTER;
constexpr int PUSH_SORT_TILE_CELLS = WXA_TILE * WXA_TILE * WXA_TILE;
// A cell's slots in the sorted tile: first the particles that an LDS-tile kernel counted in its OWN tile's histogram (rank
// from an LDS atomic; the workgroup adds its counts to the global histogram once per cell and leaves them in `own`), then
// the FOREIGN ones -- particles that changed tile, stragglers, the appended tail, every particle of a run without tiles --
// whose rank comes from a global counter per cell and carries this bit: slot = own[cell] + rank.  (As one global atomic
// per particle the counting push of 256^3 x 8 particles took 17 ms instead of 4.0; with local ranks turned into global
// ones by a second pass over the tile's record, 5.0 -- profiles/round5/README.md.)
constexpr unsigned long long PUSH_SORT_FOREIGN = 0x80000000ull;

// What the push kernels are handed (by value).  `first`: index, in the whole tile, of element 0 of the particle view the
// kernel works on (the global-memory kernel of the appended tail gets a view that starts behind the sorted part).
struct PushSort {
    int mode = 0;                                   // 0, PUSH_SORT_COUNT, PUSH_SORT_SCATTER or both
    int check_retired = 0;                          // COUNT: the tile may hold retired particles (their ids are looked at)
    double predict_dt = 0.0;                        // COUNT: key the position this many seconds of free flight ahead
    long first = 0;
    // COUNT
    SortGeom sg{};
    unsigned long long* __restrict__ kr_out = nullptr;   // (key << 32 | rank), indexed by the particle's index AFTER this push
    int* __restrict__ hist = nullptr;               // particles per key: what wxa_push_sort_end scans
    int* __restrict__ fcnt = nullptr;               // foreign particles per key so far
    int* __restrict__ own_out = nullptr;            // own particles per key (written by the tile's workgroup)
    // SCATTER
    const unsigned long long* __restrict__ kr_in = nullptr;   // indexed by the particle's index BEFORE this push
    const int* __restrict__ offs = nullptr;                  // exclusive scan of the histogram kr_in was taken with
    const int* __restrict__ own_in = nullptr;
    double* __restrict__ dx = nullptr; double* __restrict__ dy = nullptr; double* __restrict__ dz = nullptr;
    double* __restrict__ dw = nullptr;
    double* __restrict__ dux = nullptr; double* __restrict__ duy = nullptr; double* __restrict__ duz = nullptr;
    unsigned long long* __restrict__ did = nullptr;
    long np_counted = 0;      // particles kr_in covers; later ones were appended since the count
    long n_appended = 0;      // particles behind np_counted
    int retired_bin_in = 0;   // offs[retired_bin_in]: first index behind the cell-sorted particles
};

// index of particle `gi` (index in the whole tile before this push) in the destination tile; dropped: the record had it
// retired -- it lands behind the tile's new end and is not a particle of the tile any more
__device__ __forceinline__ long push_sort_dest(const PushSort& h, const long gi, bool& dropped) {
    dropped = false;
    if (gi >= h.np_counted) return (long)h.offs[h.retired_bin_in] + (gi - h.np_counted);   // behind the live ones, in their order
    const unsigned long long kr = h.kr_in[gi];
    const int key = (int)(kr >> 32);
    long d = (long)h.offs[key] + (long)(unsigned)(kr & 0x7fffffffull);
    if (kr & PUSH_SORT_FOREIGN) d += h.own_in[key];
    if (key == h.retired_bin_in) { d += h.n_appended; dropped = true; }   // the retired ones behind the appended ones: dropped by the new count
    return d;
}

// After the push of particle `ip` of view `p` (new position and momentum in registers; MOVE pushes only).
// Returns true when the caller still has to store position and momentum in place.
// lds_hist / my_tile: the LDS-tile kernel's histogram of the cells of its own tile (see push_sort_tile_finish).
__device__ __forceinline__ bool push_sort_tail(const PushSort& h, const PV& p, const long ip, const double xp, const double yp,
                                               const double zp, const double ux, const double uy, const double uz,
                                               int* lds_hist = nullptr, const long my_tile = -1) {
    if (h.mode == 0) return true;   // uniform
    const long gi = h.first + ip;
    long at = gi;                   // the particle's index after this push
    bool in_place = true;
    const bool scatter = (h.mode & PUSH_SORT_SCATTER) != 0;
    unsigned long long pid = 0ull;
    if (p.id && (scatter || h.check_retired)) pid = p.id[ip];
    bool dropped = false;
    if (scatter) {
        const double w = p.w[ip];
        at = push_sort_dest(h, gi, dropped);
        h.dx[at] = xp; h.dy[at] = yp; h.dz[at] = zp; h.dw[at] = w;
        h.dux[at] = ux; h.duy[at] = uy; h.duz[at] = uz;
        if (p.id && h.did) h.did[at] = pid;
        in_place = false;
    }
    // COUNT and SCATTER in one push: a particle the SCATTER drops (retired when the old record was taken: it lies behind
    // the new tile's end) is not in the new record -- counted, it would push the ranks of the retired bin, and with them
    // the next scatter's destinations, beyond the tile (found by the 2 x 2 x 2 bricks on the CPU execution model, round 6)
    if ((h.mode & PUSH_SORT_COUNT) && !dropped) {
        // predict_dt: the key of where the particle will be after the NEXT push if nothing accelerates it -- the push
        // that scatters then writes the particles in (all but exactly) the cell order of the positions it produces, and
        // the deposition behind it and the next gather meet a fresh sort instead of one that is a push old.  Any key
        // gives a valid order; a particle the fields deflect into another cell is one more straggler.
        double kx = xp, ky = yp, kz = zp;
        if (h.predict_dt != 0.0) {
            constexpr double inv_c2 = 1.0 / (PhysConst::c * PhysConst::c);
            const double s_ = h.predict_dt / sqrt(1.0 + (ux * ux + uy * uy + uz * uz) * inv_c2);
            kx += ux * s_; ky += uy * s_; kz += uz * s_;
        }
        const int key = (h.check_retired && p.id && pid == WXA_IDCPU_RETIRED) ? h.sg.retired_bin : cell_of(h.sg, kx, ky, kz);
        unsigned long long rank;
        if (lds_hist && key / PUSH_SORT_TILE_CELLS == my_tile) {
            rank = (unsigned long long)(unsigned)atomicAdd(&lds_hist[key % PUSH_SORT_TILE_CELLS], 1);
        } else {
            rank = PUSH_SORT_FOREIGN | (unsigned long long)(unsigned)atomicAdd(&h.fcnt[key], 1);
            atomicAdd(&h.hist[key], 1);
        }
        h.kr_out[at] = ((unsigned long long)(unsigned)key << 32) | rank;
    }
    return in_place;
}

// End of an LDS-tile kernel's workgroup in COUNT mode (every lane of the workgroup, PUSH_SORT_TILE_CELLS lanes, behind a
// barrier that follows the last push_sort_tail): the tile's own counts go to the global histogram, one atomic per
// occupied cell, and stay in `own` for the SCATTER, which puts the cell's foreign particles behind them.
__device__ __forceinline__ void push_sort_tile_finish(const PushSort& h, const int* lds_hist, const long my_tile, const int tid) {
    const int n = lds_hist[tid];
    if (n > 0) {
        atomicAdd(&h.hist[my_tile * PUSH_SORT_TILE_CELLS + tid], n);
        h.own_out[my_tile * PUSH_SORT_TILE_CELLS + tid] = n;
    }
}

// the hook of a push kernel whose particle view starts `first` particles into the tile (nothing armed, no workspace or
// a push that does not move the particles: an inert hook)
inline PushSort make_push_sort(const wxa_workspace* ws, const long first, const bool move) {
    PushSort h{};
    if (!ws || !move || ws->ps.armed == 0) return h;
    const auto& s = ws->ps;
    h.mode = s.armed;
    h.first = first;
    if (s.armed & PUSH_SORT_COUNT) {
        for (int d = 0; d < 3; ++d) { h.sg.plo[d] = s.plo[d]; h.sg.dinv[d] = s.dinv[d]; h.sg.nc[d] = s.nc[d]; h.sg.wrap[d] = s.wrap[d]; }
        h.sg.retired_bin = (int)s.bins;
        h.check_retired = s.check_retired;
        h.predict_dt = s.predict_dt;
        h.kr_out = (unsigned long long*)s.kr[s.out].p;
        h.hist = (int*)s.hist.p;
        h.fcnt = (int*)s.hist.p + (s.bins + 2);
        h.own_out = (int*)s.own[s.out].p;
    }
    if (s.armed & PUSH_SORT_SCATTER) {
        h.kr_in = (const unsigned long long*)s.kr[s.in].p;
        h.offs = (const int*)s.offs[s.in].p;
        h.own_in = (const int*)s.own[s.in].p;
        h.dx = s.dst.x; h.dy = s.dst.y; h.dz = s.dst.z; h.dw = s.dst.w;
        h.dux = s.dst.ux; h.duy = s.dst.uy; h.duz = s.dst.uz;
        h.did = (unsigned long long*)s.dst.idcpu;
        h.np_counted = s.pending_np;
        h.n_appended = s.appended;
        h.retired_bin_in = (int)s.pending_bins;
    }
    return h;
}

}  // namespace wxa
#endif
