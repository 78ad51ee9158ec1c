// Tiles that hold far more particles than the others (a wake's density spike: several 10^5 particles in a few cells of
// ONE tile of 8^3 cells) are worked on by several workgroups.
//
// The LDS-tile kernels give a tile to one workgroup; with a spike in it that workgroup runs for hundreds of milliseconds
// while the other 255 CUs have long finished (BASELINE config 5 on one GPU, 256 x 256 x 512 x 8 per cell: 277 ms per
// deposition, 40 x the uniform plasma's cost per particle -- profiles/round5/README.md).  A tile of more than
// `heavy` particles is split into k = ceil(n / heavy) units; unit 0 is the tile's regular workgroup, units 1 .. k - 1 are
// extra workgroups appended to the grid.  Every unit stages / zeroes its own LDS tile and does the tile's set-up; it takes
// every k-th chunk (deposition) or the u-th part of the particle range (gather).  The deposition's units flush their own
// LDS tiles with global atomics, which add up; the gather's units share nothing.
// A launch without a heavy tile: one tiny planning kernel, one load per workgroup.
#ifndef WXA_HEAVY_TILES_HPP_
#define WXA_HEAVY_TILES_HPP_

#include "workspace.hpp"

#include <stdlib.h>

namespace wxa {

struct HeavyUnits {
    const int* __restrict__ kt = nullptr;       // units per tile (>= 1); nullptr: every tile is one unit
    const int* __restrict__ extra = nullptr;    // (tile, unit) of the extra workgroups, two ints each
    const unsigned* __restrict__ nextra = nullptr;
    long grid_tiles = 0;                        // workgroups of the regular grid (xcd_grid_size(ntiles)); extras follow
    // streaming deposition: lanes of a wave that share a frame sum their values over the wave first when there are at
    // least this many of them (deposit_tile.hip; WXA_WAVE_SUM_MIN, read per launch; 65: never)
    int wave_sum_min = 16;
    // Tile t on workgroup t -- consecutive tiles on different XCDs -- instead of a contiguous eighth of the tiles per XCD
    // (xcd_tile_id): where the work per tile follows the plasma's structure (a wake: the dense tiles sit in a few z slabs)
    // an XCD's eighth of the domain can hold most of the launch's work, and the launch lasts as long as that XCD.
    int interleave = 0;
};

// the workgroup's tile and its share (u of k); false: nothing to do
__device__ __forceinline__ bool heavy_unit_of(const HeavyUnits& hu, const long bid, const long ntiles, long& tile, int& u, int& k) {
    u = 0; k = 1;
    if (!hu.kt || bid < hu.grid_tiles) {
        tile = hu.interleave ? bid : xcd_tile_id(bid, ntiles);
        if (tile >= ntiles) return false;
        if (hu.kt) k = hu.kt[tile];
        return true;
    }
    const long e = bid - hu.grid_tiles;
    if (e >= (long)*hu.nextra) return false;
    tile = hu.extra[2 * e];
    if (tile < 0) return false;   // an entry the planner withdrew (its overflow guard)
    u = hu.extra[2 * e + 1];
    k = hu.kt[tile];
    return true;
}

static __global__ void __launch_bounds__(256)
plan_heavy_tiles_kernel(const int* __restrict__ offsets, long ntiles, int cells_per_tile, int heavy, long max_extra,
                        int* __restrict__ kt, int* __restrict__ extra, unsigned* __restrict__ nextra) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const int n = offsets[(t + 1) * cells_per_tile] - offsets[t * cells_per_tile];
    int k = n > heavy ? (n + heavy - 1) / heavy : 1;
    if (k > 1) {
        const unsigned base = atomicAdd(nextra, (unsigned)(k - 1));
        if ((long)base + (k - 1) > max_extra) {
            // no room (cannot happen with max_extra = np / heavy + 1; kept as a guard): the tile stays whole, and the
            // entries it claimed -- as far as the table reaches -- name no tile, so that the extra workgroups that read
            // them leave at once (heavy_unit_of) instead of working on whatever the table held
            for (long e = base; e < (long)base + (k - 1) && e < max_extra; ++e) { extra[2 * e] = -1; extra[2 * e + 1] = 0; }
            k = 1;
        } else {
            for (int u = 1; u < k; ++u) { extra[2 * (base + u - 1)] = (int)t; extra[2 * (base + u - 1) + 1] = u; }
        }
    }
    kt[t] = k;
}

// Particles a tile may hold before it is split.  On for the workspaces of a streaming plasma (a boosted-frame wake is
// where the spikes were met) and wherever WXA_HEAVY_TILE says so (read per launch: a test sets it for one case; 0 switches
// the splitting off).  4096 -- the uniform plasma's tile at 8 per cell: BASELINE config 5 on one GPU, deposition per launch
// with the tiles in turn over the XCDs (profiles/round6, session za): 40.0 ms at 1024, 33.8 at 2048, 32.4 at 3072, 32.0 at
// 4096, 32.7 at 6144, 33.7 at 8192 (the default until then), 36.9 at 16384.  Off otherwise: the np / heavy + 1 extra
// workgroups a launch appends exit at once when no tile is heavy, but they cost the uniform-plasma headline 0.07 ms per step
// (profiles/round5/README.md).
inline int heavy_tile_threshold(const wxa_workspace* ws) {
    const char* e = getenv("WXA_HEAVY_TILE");
    if (e) return atoi(e);
    return ws->streaming_plasma ? 4096 : 0;
}

// Plans the units of a launch over `ntiles` tiles of `np` sorted particles; on return `hu` is what the kernel takes and
// `extra_groups` the number of workgroups to append to the regular grid.  Scratch lives in the workspace.
inline wxa_status plan_heavy_tiles(wxa_workspace* ws, const int* offsets, long ntiles, long np, HeavyUnits& hu, long& extra_groups,
                                   hipStream_t st) {
    hu = HeavyUnits{};
    extra_groups = 0;
    if (const char* e = getenv("WXA_WAVE_SUM_MIN")) hu.wave_sum_min = atoi(e);
    const int heavy = heavy_tile_threshold(ws);
    if (heavy <= 0 || np <= heavy) return WXA_OK;   // no tile can be heavy
    const long max_extra = np / heavy + 1;
    wxa_status rc;
    if ((rc = ws->heavy.reserve(sizeof(int) * (size_t)(ntiles + 2 * max_extra + 4))) != WXA_OK) return rc;
    int* kt = (int*)ws->heavy.p;
    int* extra = kt + ntiles;
    unsigned* nextra = (unsigned*)(extra + 2 * max_extra);
    WXA_HIP_CHECK(hipMemsetAsync(nextra, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(plan_heavy_tiles_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, st, offsets, ntiles,
                       WXA_TILE * WXA_TILE * WXA_TILE, heavy, max_extra, kt, extra, nextra);
    WXA_LAUNCH_CHECK();
    hu.kt = kt; hu.extra = extra; hu.nextra = nextra; hu.grid_tiles = xcd_grid_size(ntiles);
    if (const char* e = getenv("WXA_TILE_INTERLEAVE")) hu.interleave = atoi(e);   // (read per launch: the A/B of profiles/round6)
    else hu.interleave = ws->streaming_plasma ? 1 : 0;
    extra_groups = max_extra;
    return WXA_OK;
}

}  // namespace wxa
#endif
