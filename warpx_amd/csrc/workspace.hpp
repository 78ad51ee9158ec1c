// Per-device scratch owned by the caller through wxa_workspace_create/destroy.
#ifndef WXA_WORKSPACE_HPP_
#define WXA_WORKSPACE_HPP_

#include "common.hpp"
#include "shapes.hpp"

#define WXA_TILE 8   // tile edge (cells) of the tile-major cell sort and of the LDS-tile kernels

namespace wxa {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    wxa_status reserve(size_t bytes) {
        if (bytes <= cap) return WXA_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            set_last_error("hipMalloc of %zu bytes failed", want);
            return WXA_ERR_NOMEM;
        }
        cap = want;
        return WXA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace wxa

struct wxa_workspace {
    wxa::DevBuf cell, rank, hist, offsets, scan_tmp, tile_offsets, stragglers, counters;
    wxa::DevBuf heavy;   // units per tile and the extra workgroups of the tiles that are split (heavy_tiles.hpp)
    // The straggler counts of the two LDS-tile kernels live in two slots each: launch n counts in slot n & 1, which launch
    // n - 1's tile kernel left at zero, and zeroes the other one for launch n + 1 -- no fill dispatch in front of the kernel
    // (round 6: a 4-byte hipMemsetAsync is a dispatch of 5 us; wxa::flip_counter below)
    bool flip_ready = false;
    unsigned gather_flips = 0, deposit_flips = 0;
    // description of the last cell sort (consumed by the tile-based deposition)
    bool sorted_valid = false;
    int64_t sorted_np = 0;              // particles covered by the tile offsets (live ones after wxa_sort_live_count)
    int64_t sorted_bins = 0;            // number of cell bins of that sort (the retired bin follows)
    const double* sorted_x = nullptr;   // identity of the sorted particle array
    int32_t sort_nc[3] = {0, 0, 0};
    int32_t sort_cell_lo[3] = {0, 0, 0};
    double sort_plo[3] = {0, 0, 0};
    double sort_dinv[3] = {0, 0, 0};
    // particles.E_external_particle / B_external_particle of the container that owns this workspace
    // (wxa_workspace_set_external_particle_fields); added to the gathered fields in PushPX / PushP
    double ext_eb[6] = {0, 0, 0, 0, 0, 0};
    // repeated plasma lens of that container (wxa_workspace_set_repeated_plasma_lens) and the time its fields are
    // evaluated at (wxa_workspace_set_time)
    wxa::DevBuf lens_tab, ext_pp;   // ext_pp: per-particle external fields of the current push (4 x np)
    int32_t lens_n = 0;
    double lens_period = 0, lens_dt = 0, lens_gamma_boost = 1, ext_time = 0;
    int64_t ext_pp_stride = 0;
    // accumulator type of the LDS-tile Esirkepov deposition (wxa_workspace_set_deposit_accumulator)
    int32_t deposit_accumulator = WXA_ACC_FP64;
    // the container's plasma streams through the grid (wxa_workspace_set_streaming_plasma): the LDS-tile Esirkepov
    // deposition takes every particle through the wide-frame body inside its loop (deposit_tile.hip, RowsCfg::WL)
    int32_t streaming_plasma = 0;
    // the cell sort folded into PushPX (push_sort.hpp; wxa_push_sort_begin / _end): what is armed for the pushes between
    // begin and end, and the record a COUNT left for the SCATTER of a later push
    struct PushSortState {
        int32_t armed = 0;                   // WXA_PUSH_SORT_* of the pushes between begin and end
        wxa::DevBuf kr[2], offs[2], own[2], hist;   // (key, rank) per particle, the scanned histogram and the own counts
                                                    // per cell, double-buffered; hist: the histogram and the foreign counters
        int32_t check_retired = 0;           // armed COUNT: the caller's tile may hold retired particles
        double predict_dt = 0.0;             // armed COUNT: keys of the positions this much free flight ahead
        int32_t in = 0, out = 0;             // kr[in], offs[in]: the pending record; [out]: what the armed COUNT writes
        bool pending = false;
        int64_t pending_np = 0, pending_bins = 0;
        const double* pending_x = nullptr;   // identity of the tile the record indexes
        int32_t p_nc[3] = {0, 0, 0}, p_cell_lo[3] = {0, 0, 0};   // geometry of the pending record
        double p_plo[3] = {0, 0, 0}, p_dinv[3] = {0, 0, 0};
        int32_t nc[3] = {0, 0, 0}, cell_lo[3] = {0, 0, 0}, wrap[3] = {0, 0, 0};   // geometry of the armed COUNT
        double plo[3] = {0, 0, 0}, dinv[3] = {0, 0, 0};
        int64_t bins = 0;
        wxa_particle_view dst{};             // armed SCATTER: the destination tile
        int64_t np_armed = 0;                // particles of the tile being pushed
        const double* count_x = nullptr;     // ... and its identity
        int64_t appended = 0;                // armed SCATTER: particles appended since the record was taken
    } ps;
};

namespace wxa {
// words `word`, `word + 1` of ws->counters as a two-slot counter (see wxa_workspace::flip_ready)
inline wxa_status flip_counter(wxa_workspace* ws, int word, unsigned& flips, hipStream_t st, unsigned*& cur, unsigned*& next) {
    wxa_status rc;
    if ((rc = ws->counters.reserve(512)) != WXA_OK) return rc;
    if (!ws->flip_ready) {   // once per workspace
        WXA_HIP_CHECK(hipMemsetAsync(ws->counters.p, 0, 512, st));
        ws->flip_ready = true;
    }
    unsigned* base = (unsigned*)ws->counters.p + word;
    cur = base + (flips & 1u);
    next = base + ((flips + 1u) & 1u);
    ++flips;
    return WXA_OK;
}
// external fields of a push: the constants, and the per-particle ones once evaluate_particle_fields (particles.hip) has run
inline ExtEB ext_of(const wxa_workspace* ws) {
    ExtEB e{};
    if (!ws) return e;
    e.ex = ws->ext_eb[0]; e.ey = ws->ext_eb[1]; e.ez = ws->ext_eb[2];
    e.bx = ws->ext_eb[3]; e.by = ws->ext_eb[4]; e.bz = ws->ext_eb[5];
    if (ws->lens_n > 0 && ws->ext_pp_stride > 0) e.pp = ExtPerParticle{(const double*)ws->ext_pp.p, ws->ext_pp_stride};
    return e;
}
// the same for a view that starts `first` particles into the array the per-particle fields were evaluated on
inline ExtEB ext_of(const wxa_workspace* ws, int64_t first) {
    ExtEB e = ext_of(ws);
    if (e.pp.fields) e.pp.fields += first;
    return e;
}
inline ExtLens lens_of(const wxa_workspace* ws) {
    ExtLens L{};
    L.n = ws->lens_n;
    L.period = ws->lens_period;
    L.time = ws->ext_time;
    L.dt = ws->lens_dt;
    L.gamma_boost = ws->lens_gamma_boost;
    // m_uz_boost (GetExternalFields.cpp:28)
    L.uz_boost = std::sqrt(ws->lens_gamma_boost * ws->lens_gamma_boost - 1.0) * PhysConst::c;
    L.tab = (const double*)ws->lens_tab.p;
    return L;
}
// LDS-tile deposition (deposit_tile.hip)
bool deposit_tile_available(const wxa_workspace* ws, const wxa_particle_view* p);
wxa_status deposit_current_tiled(const wxa_particle_view* p, const wxa_field_view J[3],
                                 const wxa_grid_geom* geom, double q, double dt, double relative_time,
                                 int order, int algo, wxa_workspace* ws, hipStream_t stream);
// LDS-tile gather + push (gather_tile.hip)
bool gather_tile_available(const wxa_workspace* ws, const wxa_particle_view* p);
wxa_status gather_push_tiled(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                             const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                             int pusher, bool move, wxa_workspace* ws, hipStream_t stream);
wxa_status gather_push_tiled_part(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                                  const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                                  int pusher, int part, wxa_workspace* ws, hipStream_t stream);
}  // namespace wxa
#endif
